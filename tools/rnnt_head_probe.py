"""A few launches of the transducer output layer's two fused kernels at the config-5 sizes (J = 512, V = 1000): the forward
reduction over all cells of a micro-batch and the gradient of one 65 536-cell chunk; for counter passes and kernel traces."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from emoasr_amd import ops

dev = torch.device("cuda:0")
B, T, U, V, J = 22, 320, 40, 1000, 512
torch.manual_seed(0)
h = torch.tanh(torch.randn(B * T * U, J, device=dev)).to(torch.bfloat16)
w = (torch.randn(V, J, device=dev) / J ** 0.5).to(torch.bfloat16)
bias = torch.zeros(V, device=dev)
labels = torch.randint(1, V, (B, U - 1), device=dev, dtype=torch.int32)
elens = torch.full((B,), T, device=dev, dtype=torch.int32)
ylens = torch.full((B,), U - 1, device=dev, dtype=torch.int32)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
CH = 65536
Vp = (V + 63) // 64 * 64
dz = torch.empty(CH, Vp, device=dev, dtype=torch.bfloat16)[:, :V]
for _ in range(n):
    ctx, nll = ops.rnnt_head_forward(h, w, bias, B, T, U, labels, elens, ylens, 0)
    coef, ycol = ops.rnnt_coef(ctx, nll, labels, elens, ylens, 1.0 / B)
    for r0 in range(0, 2 * CH, CH):
        ops.rnnt_head_grad(h[r0:r0 + CH], w, bias, coef[r0:r0 + CH], ycol[r0:r0 + CH], 0, dz)
torch.cuda.synchronize()
