"""A few launches of ONE product of the stacked training step (M = 35 145 rows) for counter passes (rocprofv3 --pmc) and kernel
traces: python tools/gemm_probe.py <ffn1|ffn1_epi|ffn2|d_ffn2_epi|qkv|out_epi> [launches]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from emoasr_amd import ops

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "ffn1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
M = int(os.environ.get("M", 35145))
dt = torch.bfloat16
rnd = lambda *s: torch.randn(*s, device=dev).to(dt)
if which.startswith("ffn1"):
    a, b, bias = rnd(M, 256), rnd(1024, 256), torch.randn(1024, device=dev)
    out, pre = torch.empty(M, 1024, device=dev, dtype=dt), torch.empty(M, 1024, device=dev, dtype=dt)
    f = (lambda: ops.gemm_nt(a, b, out=out, bias=bias, act=ops.ACT_SWISH, pre_out=pre, drop_p=0.1, seed=5)) if which.endswith("epi") \
        else (lambda: ops.gemm_nt(a, b, out=out, bias=bias))
elif which == "ffn2":
    a, b, bias, x = rnd(M, 1024), rnd(256, 1024), torch.randn(256, device=dev), rnd(M, 256)
    f = lambda: ops.gemm_nt(a, b, out=x, bias=bias, residual=x, res_scale=0.5, drop_p=0.1, seed=5)
elif which == "d_ffn2_epi":
    a, b, pre = rnd(M, 256), rnd(256, 1024), rnd(M, 1024)
    out = torch.empty(M, 1024, device=dev, dtype=dt)
    f = lambda: ops.gemm_nn(a, b, out=out, dact_pre=pre, dact=ops.ACT_SWISH, drop_p=0.1, seed=5)
elif which == "qkv":
    a, b, bias = rnd(M, 256), rnd(768, 256), torch.randn(768, device=dev)
    out = torch.empty(M, 768, device=dev, dtype=dt)
    f = lambda: ops.gemm_nt(a, b, out=out, bias=bias)
else:
    a, b, bias, x = rnd(M, 256), rnd(256, 256), torch.randn(256, device=dev), rnd(M, 256)
    f = lambda: ops.gemm_nt(a, b, out=x, bias=bias, residual=x, res_scale=1.0, drop_p=0.1, seed=5)
for _ in range(n):
    f()
torch.cuda.synchronize()
