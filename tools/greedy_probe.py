"""Batch-1 greedy CTC decoding of the L2 model (bf16): wall time per utterance; run under `rocprofv3 --kernel-trace` for the
kernel list (tools/kstats.py) of the 20 timed utterances."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
import bench
from emoasr_amd.hostenv import respect_cpu_quota
from emoasr_amd.modeling.asr import ASR
respect_cpu_quota()
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ASR(SimpleNamespace(**bench.L2), compute_dtype=torch.bfloat16).to(dev).eval()
g = torch.Generator().manual_seed(1)
utts = [torch.randn(1, 1200, 80, generator=g).to(dev) for _ in range(4)]
with torch.no_grad():
    for x in utts: model.decode(x, [1200])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for i in range(n): model.decode(utts[i % 4], [1200])
    torch.cuda.synchronize()
print(f"greedy decode, 12 s utterances: {1e3 * (time.perf_counter() - t0) / n:.3f} ms per utterance")
