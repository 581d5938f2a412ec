import os, sys, cProfile, pstats, io
sys.path.insert(0, '/root/repo')
from types import SimpleNamespace
import torch
import bench
from emoasr_amd.hostenv import respect_cpu_quota
from emoasr_amd.modeling.asr import ASR
respect_cpu_quota()
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ASR(SimpleNamespace(**bench.L2), compute_dtype=torch.bfloat16).to(dev).eval()
x = torch.randn(1, 1200, 80).to(dev)
with torch.no_grad():
    for _ in range(5): model.decode(x, [1200])
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(50): model.decode(x, [1200])
    torch.cuda.synchronize()
    pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:5000])
