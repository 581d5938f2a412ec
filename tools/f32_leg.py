"""The f32 (parity) mode's training step alone -- bench.parity_mode's timing part: frames/s and ms per step of one 27 k-frame batch
per optimizer step through the exact-f32 kernels (not stacked)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from emoasr_amd.hostenv import respect_cpu_quota

respect_cpu_quota()
dev = torch.device("cuda:0")
batches = bench.make_batches(0, 1, 10, dev)
import time  # noqa: E402
from types import SimpleNamespace  # noqa: E402

from emoasr_amd.modeling.asr import ASR  # noqa: E402
from emoasr_amd.train import ArenaAdam, noam_lr  # noqa: E402

torch.manual_seed(0)
MODE = "f32x3" if "--split" in sys.argv else torch.float32   # --split: f32 storage, split-bf16 products
m32 = ASR(SimpleNamespace(**bench.L2), compute_dtype=MODE).to(dev).train()
opt = ArenaAdam(m32.engine().arena, lambda s: noam_lr(bench.OPT["lr"], 256, bench.OPT["warmup"], s),
                weight_decay=bench.OPT["weight_decay"], clip_grad_norm=bench.OPT["clip_grad_norm"])


def step(bt):
    loss, _ = m32(bt.xs, bt.xlens, bt.ys, bt.ylens, None, None)
    opt.zero_grad()
    loss.backward()
    opt.step()


def step_stacked(group):
    opt.zero_grad()
    m32.engine().ctc_train_stacked([(bt.xs, bt.xlens, bt.ys, bt.ylens) for bt in group], bench.L2["blank_id"])
    opt.step()


if "--stacked" in sys.argv:   # five micro-batches per optimizer step in one stacked pass (9 steps, 3 of them warm-up)
    batches = bench.make_batches(0, 1, 45, dev)
    groups = [batches[5 * i:5 * i + 5] for i in range(9)]
    for g in groups[:3]:
        step_stacked(g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for g in groups[3:9]:
        step_stacked(g)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    nfr = sum(sum(b.xlens) for g in groups[3:9] for b in g)
    print(f"{'f32x3' if MODE == 'f32x3' else 'f32'} stacked x5: {nfr / el:.0f} frames/s, {1e3 * el / 6:.2f} ms per step")
    sys.exit(0)
for bt in batches[:3]:
    step(bt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for bt in batches[3:9]:
    step(bt)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"{'f32x3' if MODE == 'f32x3' else 'f32'}: {sum(sum(b.xlens) for b in batches[3:9]) / el:.0f} frames/s, {1e3 * el / 6:.2f} ms per step")
