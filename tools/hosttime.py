"""Measure host-side cost of one training step: (a) python only (C calls skipped), (b) full."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
import bench
from emoasr_amd import lib as emo_lib, ops
from emoasr_amd.modeling.asr import ASR
from emoasr_amd.train import ArenaAdam, noam_lr

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ASR(SimpleNamespace(**bench.L2), compute_dtype=torch.bfloat16).to(dev).train()
eng = model.engine()
opt = ArenaAdam(eng.arena, lambda s: noam_lr(5.0, 256, 25000, s), weight_decay=1e-6, clip_grad_norm=5.0)
batches = bench.make_batches(0, 1, 12, dev)

def step(bt):
    loss, _ = model(bt.xs, bt.xlens, bt.ys, bt.ylens, None, None)
    opt.zero_grad(); loss.backward(); opt.step()

for bt in batches[:3]: step(bt)
torch.cuda.synchronize()
# full
t0 = time.perf_counter()
for bt in batches[3:9]: step(bt)
t_launch = time.perf_counter() - t0
torch.cuda.synchronize()
t_full = time.perf_counter() - t0
print(f"full: launch-return {1e3*t_launch/6:.2f} ms/step, with sync {1e3*t_full/6:.2f} ms/step")
# count calls
n = [0]
orig = emo_lib.call
def counting(name, *a):
    n[0] += 1
    return orig(name, *a)
emo_lib.call = counting; ops.lib.call = counting
step(batches[9]); torch.cuda.synchronize()
print("C calls per step:", n[0])
# python only
def dry(name, *a): return None
emo_lib.call = dry; ops.lib.call = dry
t0 = time.perf_counter()
for bt in batches[3:9]: step(bt)
torch.cuda.synchronize()
print(f"python-only (C calls skipped): {1e3*(time.perf_counter()-t0)/6:.2f} ms/step")
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for bt in batches[3:9]: step(bt)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr, stream=sys.stdout).sort_stats("tottime").print_stats(28)
