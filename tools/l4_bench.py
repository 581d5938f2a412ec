"""config 5 (RNN-T) training step alone at a given sampler budget: python tools/l4_bench.py [frames] [ylens]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import random
from types import SimpleNamespace

import torch

import bench
from emoasr_amd.data import libri_shaped_lengths, pack_batches
from emoasr_amd.modeling.asr import ASR
from emoasr_amd.train import ArenaAdam, noam_lr

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
ylb = int(sys.argv[2]) if len(sys.argv) > 2 else frames // 10
steps, warmup = 6, 3
dev = torch.device("cuda:0")
torch.manual_seed(2)
model = ASR(SimpleNamespace(**bench.L4), compute_dtype=torch.bfloat16).to(dev).train()
eng = model.engine()
opt = ArenaAdam(eng.arena, lambda s: noam_lr(bench.OPT["lr"], 256, bench.OPT["warmup"], s), weight_decay=bench.OPT["weight_decay"],
                clip_grad_norm=bench.OPT["clip_grad_norm"])
xlens, ylens = libri_shaped_lengths(2000, 0)
batches = pack_batches(xlens, ylens, frames, ylb, 50, 1)
random.Random(3).shuffle(batches)
g = torch.Generator().manual_seed(5)
data = []
for idx in batches[: steps + warmup]:
    xl, yl = [int(xlens[i]) for i in idx], [int(ylens[i]) for i in idx]
    xs = torch.randn(len(idx), max(xl), 80, generator=g)
    ys = torch.randint(3, bench.L4["vocab_size"], (len(idx), max(yl)), generator=g)
    eos = torch.full((len(idx), 1), bench.L4["eos_id"])
    for b in range(len(idx)):
        xs[b, xl[b]:] = 0
        ys[b, yl[b]:] = bench.L4["eos_id"]
    data.append((xs.to(dev), xl, ys, yl, torch.cat([eos, ys], 1), torch.cat([ys, eos], 1)))
    print(f"batch B {len(idx)} T {max(xl)} T' {((max(xl) - 1) // 2 - 1) // 2} U {max(yl) + 1}", flush=True)


def step(bt):
    loss, _ = model(*bt)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss


for bt in data[:warmup]:
    step(bt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for bt in data[warmup:]:
    loss = step(bt)
torch.cuda.synchronize()
el = time.perf_counter() - t0
fr = sum(sum(bt[1]) for bt in data[warmup:])
print(f"budget {frames}: {fr / el / 1e6:.3f} M frames/s  {1e3 * el / steps:.2f} ms/step  loss {float(loss):.3f}  "
      f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
