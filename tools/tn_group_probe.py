"""time one encoder layer's grouped weight-gradient launch for several split-K block targets"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emoasr_amd import lib, ops
dev = torch.device("cuda:0")
M = int(os.environ.get("M", 6840))
dt = torch.bfloat16
shapes = [(1024, 256), (256, 1024), (1024, 256), (256, 1024), (768, 256), (256, 256), (512, 256), (256, 256)]  # (N1, N2)
probs = []
for n1, n2 in shapes:
    a = torch.randn(M, n1, device=dev).to(dt); b = torch.randn(M, n2, device=dev).to(dt)
    probs.append((a, b, torch.zeros(n1, n2, device=dev), 1.0, torch.zeros(n1, device=dev), 1.0))
fl = sum(2.0 * M * n1 * n2 for n1, n2 in shapes)
for blocks in (96, 128, 192, 256, 384, 512, 768):
    lib.set_option("tn_group_blocks", blocks)
    for _ in range(5): ops.gemm_tn_grouped(probs)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(30): ops.gemm_tn_grouped(probs)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 30 * 1e3
    print(f"target {blocks:5d} blocks: {us:7.1f} us  {fl / us * 1e-6:6.1f} TF/s", flush=True)
