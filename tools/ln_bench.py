"""Device time of the LayerNorm passes at the stacked training shape (M = 35 145 rows of 256 features, bf16; HIP-graph timed):
the half-wave-per-row forward kernel (option ln_fwd8) against the one-wave-per-row kernel, and the backward as the step calls it
(residual gradient in, dropped-out branch gradient out, deferred dgamma / dbeta fold)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from emoasr_amd import lib, ops
from tools._timing import graph_time

dev = torch.device("cuda:0")
M, N = int(os.environ.get("M", 35145)), int(os.environ.get("N", 256))
for dtype in (torch.bfloat16, torch.float32):
    x = torch.randn(M, N, device=dev).to(dtype)
    dy, dres = torch.randn_like(x), torch.randn_like(x)
    gamma, beta = 1 + 0.1 * torch.randn(N, device=dev), 0.1 * torch.randn(N, device=dev)
    dg, db = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    sz = x.element_size()
    ys = []
    for opt in (0, 1):
        lib.set_option("ln_fwd8", opt)
        us = graph_time(lambda: ops.layernorm_fwd(x, gamma, beta, 1e-5))
        ys.append(ops.layernorm_fwd(x, gamma, beta, 1e-5))
        print(f"{str(dtype)[6:]:9s} fwd ln_fwd8={opt}: {us:7.2f} us  {2 * M * N * sz / us / 1e6:6.2f} TB/s (x in, y out)")
    d = [(a.float() - b.float()).abs().max().item() for a, b in zip(ys[0], ys[1])]
    print(f"          max |difference| between the two forward kernels: y {d[0]:.2e}  mean {d[1]:.2e}  rstd {d[2]:.2e}")
    y, mean, rstd = ys[1]
    deferred = []
    us = graph_time(lambda: (ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres, dg, db, deferred=deferred), deferred.clear()))
    print(f"          bwd (dy, x, dres in; dx out): {us:7.2f} us  {4 * M * N * sz / us / 1e6:6.2f} TB/s")
    us = graph_time(lambda: (ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres, dg, db, branch=(0.5, 0.1, 7), deferred=deferred),
                             deferred.clear()))
    print(f"          bwd + branch gradient out   : {us:7.2f} us  {5 * M * N * sz / us / 1e6:6.2f} TB/s")
