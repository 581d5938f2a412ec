"""Multi-GPU readiness on ONE GPU: the data-parallel step under a real RCCL process group of world size 1.
The overlapped gradient buckets (train.GradBuckets driven by the engine's per-layer hook) must give exactly the gradients
of the hook-less run, and every kernel the engine enqueues during the step must go to torch's current stream -- the
stream the asynchronous all-reduces are ordered against (the weight-gradient side-stream option would break that)."""
import os
import socket
from types import SimpleNamespace

import pytest
import torch

from tests.util import CONFIGS, load_golden

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_grad_buckets_under_rccl_world1(dev, dtype, monkeypatch):
    import torch.distributed as dist
    from emoasr_amd import lib, ops
    from emoasr_amd.modeling.asr import ASR
    from emoasr_amd.train import GradBuckets
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        cfg, sd, g = load_golden("l2_tiny")
        model = ASR(SimpleNamespace(**CONFIGS["l2_tiny"]), compute_dtype=dtype)
        model.load_state_dict(sd)
        model = model.to(dev).train()
        eng = model.engine()
        assert not eng._side_wgrads  # the side stream is incompatible with the hook's stream ordering
        streams = set()
        orig = lib.call

        def recording(name, *args):
            for a in args[-1:]:
                if isinstance(a, ops.c_void_p):
                    streams.add(a.value or 0)
            return orig(name, *args)

        def run(with_hook):
            eng.step_count = 5
            eng.arena.grad.zero_()
            buckets = None
            if with_hook:
                buckets = GradBuckets(eng.arena.grad, min_elems=1 << 12)
                eng.grad_hook = buckets.ready
            loss, _ = model(g["xs"].to(dev), g["xlens"], g["ys"], g["ylens"], g["ys_in"], g["ys_out"])
            loss.backward()
            n = len(buckets.handles) if buckets is not None else 0
            if buckets is not None:
                buckets.finish()
            eng.grad_hook = None
            torch.cuda.synchronize()
            return eng.arena.grad.clone(), n

        plain, _ = run(False)
        monkeypatch.setattr(lib, "call", recording)
        monkeypatch.setattr(ops.lib, "call", recording)
        hooked, n_async = run(True)
        assert n_async >= 1  # ranges really went out during the backward sweep
        assert streams == {torch.cuda.current_stream().cuda_stream}, streams
        gmax = plain.abs().max()
        assert gmax > 0 and ((hooked - plain).abs().max() / gmax).item() < 1e-5  # (world 1: the sum is the identity)
    finally:
        dist.destroy_process_group()
