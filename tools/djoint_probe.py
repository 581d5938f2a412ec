"""The transducer joint's data gradient of one 65 536-cell chunk, dpre = (dz . W_out) * (1 - h^2): the NN product on the 64 x 64
kernel against the NT form over zero-padded columns on the large-tile kernel (HIP-graph timed)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from emoasr_amd import lib, ops
from tools._timing import graph_time

dev = torch.device("cuda:0")
M, V, J = 65536, 1000, 512
Vp = 1024
dt = torch.bfloat16
dzp = torch.zeros(M, Vp, device=dev, dtype=dt)
dzp[:, :V] = (torch.randn(M, V, device=dev) * 0.1).to(dt)
w = (torch.randn(V, J, device=dev) / J ** 0.5).to(dt)
w_t = torch.zeros(J, Vp, device=dev, dtype=dt)
w_t[:, :V].copy_(w.t())
h = torch.tanh(torch.randn(M, J, device=dev)).to(dt)
out0, out1 = torch.empty(M, J, device=dev, dtype=dt), torch.empty(M, J, device=dev, dtype=dt)
t0 = graph_time(lambda: ops.gemm_nn(dzp[:, :V], w, out=out0, dact_pre=h, dact=ops.DACT_TANH_OUT))
print(f"nn 64x64 (K = 1000): {t0:7.1f} us")
for bm in (0, 128, 192, 256):
    lib.set_option("big_bm", bm)
    t1 = graph_time(lambda: ops.gemm_nt(dzp, w_t, out=out1, dact_pre=h, dact=ops.DACT_TANH_OUT))
    print(f"nt large tile bm={bm} (K = 1024): {t1:7.1f} us   max |diff| {(out0.float() - out1.float()).abs().max().item():.2e}")
lib.set_option("big_bm", 0)
lib.set_option("big_n256", 0)
t2 = graph_time(lambda: ops.gemm_nt(dzp, w_t, out=out1, dact_pre=h, dact=ops.DACT_TANH_OUT))
print(f"nt 64x64 (K = 1024): {t2:7.1f} us")
print(f"blas: {graph_time(lambda: torch.mm(dzp, w_t.t(), out=out1)):7.1f} us")
