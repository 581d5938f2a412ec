"""the L4 (RNN-T) training step alone, for rocprofv3 --kernel-trace: bench.l4_rnnt without its decode legs"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
import bench
from emoasr_amd.data import libri_shaped_lengths, pack_batches
from emoasr_amd.modeling.asr import ASR
from emoasr_amd.train import ArenaAdam, noam_lr
dev = torch.device("cuda:0")
torch.manual_seed(2)
model = ASR(SimpleNamespace(**bench.L4), compute_dtype=torch.bfloat16).to(dev).train()
eng = model.engine()
opt = ArenaAdam(eng.arena, lambda s: noam_lr(5.0, 256, 25000, s), weight_decay=1e-6, clip_grad_norm=5.0)
xlens, ylens = libri_shaped_lengths(2000, 0)
batches = pack_batches(xlens, ylens, 12000, 1200, 50, 1)
random.Random(3).shuffle(batches)
g = torch.Generator().manual_seed(5)
data = []
for idx in batches[:10]:
    xl, yl = [int(xlens[i]) for i in idx], [int(ylens[i]) for i in idx]
    xs = torch.randn(len(idx), max(xl), 80, generator=g)
    ys = torch.randint(3, 1000, (len(idx), max(yl)), generator=g)
    eos = torch.full((len(idx), 1), 2)
    for b in range(len(idx)):
        xs[b, xl[b]:] = 0
        ys[b, yl[b]:] = 2
    data.append((xs.to(dev), xl, ys, yl, torch.cat([eos, ys], 1), torch.cat([ys, eos], 1)))
def step(bt):
    loss, _ = model(*bt)
    opt.zero_grad(); loss.backward(); opt.step()
for bt in data[:2]: step(bt)
torch.cuda.synchronize(); t0 = time.perf_counter()
for bt in data[2:]: step(bt)
torch.cuda.synchronize()
print(f"L4 train: {1e3 * (time.perf_counter() - t0) / 8:.3f} ms/step")
