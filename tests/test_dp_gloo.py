"""Data-parallel path on CPU: world_size 2, gloo.  Each rank computes gradients of ITS shard with
the oracle, the flat gradient buffer goes through emoasr_amd.train.allreduce_sum_ (the same call
the GPU trainer makes over RCCL) and is scaled by 1/world; the result must equal the gradient of
the mean of the per-replica batch-mean losses, which is what nn.DataParallel computes in the
reference (asr/train_asr.py:67-71,236-243)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import model as om
from tests.util import load_golden


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _shard(g, rank, world):
    B = g["xs"].shape[0]
    per = (B + world - 1) // world  # DataParallel scatter: contiguous chunks of ceil(B/G)
    sl = slice(rank * per, min(B, (rank + 1) * per))
    return g["xs"][sl], g["xlens"][sl], g["ys"][sl], g["ylens"][sl]


def _flat_grads(sd, cfg, batch):
    sd = {k: v.clone() for k, v in sd.items()}
    names = [k for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k]
    for k in names:
        sd[k].requires_grad_(True)
    loss, _, _ = om.asr_ctc_forward(sd, cfg, *batch, training=True)
    loss.backward()
    return loss.detach(), torch.cat([sd[k].grad.flatten() for k in names])


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from emoasr_amd.train import allreduce_sum_
    cfg, sd, g = load_golden("l1_tiny")
    loss, flat = _flat_grads(sd, cfg, _shard(g, rank, world))
    allreduce_sum_(flat)
    flat /= world
    losses = [torch.zeros(()) for _ in range(world)]
    dist.all_gather(losses, loss)
    if rank == 0:
        torch.save({"grad": flat, "losses": torch.stack(losses)}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_dp2_matches_mean_of_replica_losses(tmp_path):
    world, out = 2, str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = torch.load(out)
    cfg, sd, g = load_golden("l1_tiny")
    # single process: mean over replicas of each replica's (sum_b nll_b / B_local)
    sd = {k: v.clone() for k, v in sd.items()}
    names = [k for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k]
    for k in names:
        sd[k].requires_grad_(True)
    losses = [om.asr_ctc_forward(sd, cfg, *_shard(g, r, world), training=True)[0] for r in range(world)]
    torch.stack(losses).mean().backward()
    ref = torch.cat([sd[k].grad.flatten() for k in names])
    assert torch.allclose(got["losses"], torch.stack([l.detach() for l in losses]), rtol=1e-5)
    err = (got["grad"] - ref).abs().max() / ref.abs().max()
    assert err < 1e-5, err


def test_rank_sharding_is_disjoint_and_deterministic():
    import bench
    dev = torch.device("cpu")
    a = bench.make_batches(0, 2, 3, dev)
    b = bench.make_batches(1, 2, 3, dev)
    a2 = bench.make_batches(0, 2, 3, dev)
    for x, y, z in zip(a, b, a2):
        assert x.xlens != y.xlens or not torch.equal(x.xs, y.xs)
        assert x.xlens == z.xlens and torch.equal(x.xs, z.xs)
    assert all(sum(x.xlens) <= 30000 + max(x.xlens) for x in a)


def test_sampler_rule_and_specaug_spans():
    import numpy as np
    from emoasr_amd.data import pack_batches, specaug_spans
    xl = np.array([100, 200, 300, 400, 500])
    yl = np.array([3, 6, 10, 13, 16])
    assert pack_batches(xl, yl, max_xlens_batch=600, max_ylens_batch=100, batch_size=50) == [[0, 1, 2], [3], [4]]
    assert pack_batches(xl, yl, max_xlens_batch=10 ** 6, max_ylens_batch=100, batch_size=2) == [[0, 1], [2, 3], [4]]
    sp = specaug_spans([300, 50], 80, np_rng=np.random.RandomState(0))
    assert sp.shape == (2, 4, 2)
    assert (sp[:, :2, 1] <= 80).all() and (sp[:, :2, 1] - sp[:, :2, 0] < 30).all()
    assert (sp[0, 2:, 1] <= 300).all() and (sp[1, 2:, 1] <= 50).all()
    assert (sp[:, 2:, 1] - sp[:, 2:, 0] < 40).all()


def test_adaptive_specaug_follows_the_reference_draws():
    """asr/spec_augment.py:63-95 with max_mask_time_ratio / num_masks_time_ratio, restated literally on a numpy array and
    compared with the bands data.SpecAugment samples from identically seeded generators."""
    import random
    from types import SimpleNamespace

    import numpy as np
    from emoasr_amd.data import SpecAugment
    P = SimpleNamespace(max_mask_freq=30, num_masks_freq=2, max_mask_time_ratio=0.05, num_masks_time_ratio=0.01,
                        replace_with_zero=True)
    xlens = [1234, 310, 777]
    aug = SpecAugment(P, np_rng=np.random.RandomState(5), py_rng=random.Random(5))
    sp = aug.spans(xlens, 80)
    assert sp.shape == (3, 2 + 20, 2)
    np_rng, py_rng = np.random.RandomState(5), random.Random(5)
    for b, xlen in enumerate(xlens):
        x = np.ones((xlen, 80), np.float32)
        for f, w in np_rng.randint(0, 30, size=(2, 2)):
            f0 = py_rng.randrange(0, 80 - f)
            if f:
                x[:, f0:f0 + w] = 0
        mmt, nmt = min(20, round(xlen * 0.05)), min(20, round(xlen * 0.01))
        assert nmt == [12, 3, 8][b] and mmt == [20, 16, 20][b]
        for t, w in np_rng.randint(0, mmt, size=(nmt, 2)):
            t0 = py_rng.randrange(0, xlen - t)
            if t:
                x[t0:t0 + w] = 0
        got = np.ones((xlen, 80), np.float32)
        for m, (lo, hi) in enumerate(sp[b]):
            if m < 2:
                got[:, lo:hi] = 0
            else:
                got[lo:hi] = 0
        assert (got == x).all()
        assert (sp[b, 2 + nmt:] == 0).all()


def _bucket_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emoasr_amd.train import GradBuckets, allreduce_sum_
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(10_000, generator=g)
    ref = flat.clone()
    allreduce_sum_(ref)
    buckets = GradBuckets(flat, min_elems=1500)
    for lo in (9_500, 9_000, 7_000, 6_900, 3_000, 2_999):  # the backward sweep reports ranges from the end
        buckets.ready(lo)
    n_async = len(buckets.handles)
    buckets.finish()
    if rank == 0:
        torch.save({"same": bool(torch.equal(flat, ref)), "n_async": n_async, "done": buckets.done}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bucketed_overlap_allreduce_equals_one_allreduce(tmp_path):
    """train.GradBuckets: tail ranges all-reduced asynchronously as the backward sweep reports them sum to
    exactly one all-reduce of the whole flat gradient; small ranges are merged up to min_elems"""
    world, out = 2, str(tmp_path / "b.pt")
    mp.spawn(_bucket_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = torch.load(out)
    assert got["same"] and got["done"] == 10_000
    assert got["n_async"] == 2  # [7000,10000) and [3000,7000): the 500/1000/100/1-element pieces were merged


def _bf16_bucket_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emoasr_amd.train import GradBuckets, allreduce_sum_
    g = torch.Generator().manual_seed(200 + rank)
    flat = torch.randn(6_000, generator=g)
    ref = flat.clone()
    allreduce_sum_(ref)
    buckets = GradBuckets(flat, min_elems=1000, comm_dtype=torch.bfloat16)
    for lo in (4_000, 1_500):
        buckets.ready(lo)
    buckets.finish()
    if rank == 0:
        torch.save({"err": float((flat - ref).abs().max()), "scale": float(ref.abs().max()), "dtype": str(flat.dtype)}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bf16_gradient_buckets(tmp_path):
    """comm_dtype=bf16: half the bytes on the wire, the f32 arena receives the widened sums (bf16 rounding of the operands
    and of the sum: 2^-8 relative)"""
    world, out = 2, str(tmp_path / "b16.pt")
    mp.spawn(_bf16_bucket_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = torch.load(out)
    assert got["dtype"] == "torch.float32" and got["err"] < 2e-2 * got["scale"]
