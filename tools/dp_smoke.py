"""single-rank exercise of the data-parallel code path on a GPU box: RCCL process group of size 1, the bucketed
overlapped all-reduce hooked into the backward sweep, one optimizer step (what bench.py does for --gpus N)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
from types import SimpleNamespace
import bench
from emoasr_amd.modeling.asr import ASR
from emoasr_amd.train import ArenaAdam, GradBuckets, noam_lr
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
model = ASR(SimpleNamespace(**bench.L2), compute_dtype=torch.bfloat16).to(dev).train()
eng = model.engine()
opt = ArenaAdam(eng.arena, lambda s: noam_lr(5.0, 256, 25000, s), weight_decay=1e-6, clip_grad_norm=5.0)
buckets = GradBuckets(eng.arena.grad)
eng.grad_hook = buckets.ready
for bt in bench.make_batches(0, 1, 3, dev):
    loss, _ = model(bt.xs, bt.xlens, bt.ys, bt.ylens, None, None)
    opt.zero_grad(); loss.backward()
    n = len(buckets.handles)
    buckets.finish(); opt.step(grad_mult=1.0)
    torch.cuda.synchronize()
    print("loss", float(loss), "async buckets during backward", n, "grad finite", bool(torch.isfinite(eng.arena.grad).all()))
dist.barrier(); dist.destroy_process_group()
print("dp smoke ok")
