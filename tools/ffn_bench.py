"""Device time of the fused feed-forward block against LayerNorm + two GEMMs at the L2 shape (HIP-graph timed)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from emoasr_amd import ops
from tools._timing import graph_time

dev = torch.device("cuda:0")
M, d, F = int(os.environ.get("M", 7029)), 256, 1024
dt = torch.bfloat16
p = float(os.environ.get("P", 0.1))
x = torch.randn(M, d, device=dev).to(dt)
ln_g, ln_b = torch.ones(d, device=dev), torch.zeros(d, device=dev)
w1, b1 = (torch.randn(F, d, device=dev) / 16).to(dt), torch.zeros(F, device=dev)
w2, b2 = (torch.randn(d, F, device=dev) / 32).to(dt), torch.zeros(d, device=dev)
u = torch.empty(M, F, device=dev, dtype=dt)


def unfused():
    h, mean, rstd = ops.layernorm_fwd(x, ln_g, ln_b, 1e-5, True)
    a = ops.gemm_nt(h, w1, bias=b1, act=ops.ACT_SWISH, pre_out=u, drop_p=p, seed=11)
    return ops.gemm_nt(a, w2, bias=b2, residual=x, res_scale=0.5, drop_p=p, seed=12)


def fused():
    return ops.ffn_fwd(x, ln_g, ln_b, 1e-5, w1, b1, w2, b2, ops.ACT_SWISH, 0.5, p, 11, 12)


if os.environ.get("ONLY") != "fused":
    print(f"unfused (LN + 2 GEMM): {graph_time(unfused, n=10):7.1f} us")
print(f"fused                : {graph_time(fused, n=10):7.1f} us   ({4.0 * M * d * F / graph_time(fused, n=10) / 1e6:.0f} TF/s)")
