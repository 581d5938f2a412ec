"""The large-tile kernel's two wave layouts (option big_waves: 8 waves of (BM/2) x 64, 4 waves of (BM/2) x 128) on the products it
carries: Conv2d forward / data gradient, the CTC vocabulary head with its soft-max partials, the transducer head (forward reduction,
one gradient chunk), the joint's data gradient, the long reductions onto one column tile.  HIP-graph timed."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from emoasr_amd import lib, ops
from tools._timing import graph_time

dev = torch.device("cuda:0")
dt = torch.bfloat16
torch.manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev).to(dt)
C = 256
x = torch.randn(23, 1200, 80, device=dev)
y1 = ops.conv1_fwd(x, torch.randn(C, 9, device=dev) * 0.3, torch.randn(C, device=dev) * 0.1, dt)
w2 = (torch.randn(C, C, 3, 3, device=dev) * (9 * C) ** -0.5).to(dt)
w2p, wt = w2.permute(0, 2, 3, 1).reshape(C, 9 * C).contiguous(), w2.permute(1, 2, 3, 0).reshape(C, 9 * C).contiguous()
b2 = torch.randn(C, device=dev) * 0.1
y2 = ops.conv2_fwd(y1, w2p, bias=b2, act=ops.ACT_RELU)
dy2 = torch.randn_like(y2)
Mh = 35145
eo, wh, bh = rnd(Mh, 256), (torch.randn(10000, 256, device=dev) / 16).to(dt), torch.zeros(10000, device=dev)
B, T, U, V, J = 22, 320, 40, 1000, 512
h = torch.tanh(torch.randn(B * T * U, J, device=dev)).to(dt)
wo, bo = (torch.randn(V, J, device=dev) / J ** 0.5).to(dt), torch.zeros(V, device=dev)
labels = torch.randint(1, V, (B, U - 1), device=dev, dtype=torch.int32)
elens, ylens = torch.full((B,), T, device=dev, dtype=torch.int32), torch.full((B,), U - 1, device=dev, dtype=torch.int32)
ctx, nll = ops.rnnt_head_forward(h, wo, bo, B, T, U, labels, elens, ylens, 0)
coef, ycol = ops.rnnt_coef(ctx, nll, labels, elens, ylens, 1.0 / B)
CH = 65536
dzp = torch.zeros(CH, 1024, device=dev, dtype=dt)
w_t = torch.zeros(J, 1024, device=dev, dtype=dt)
w_t[:, :V].copy_(wo.t())
dpre = torch.empty(CH, J, device=dev, dtype=dt)
a2, wf2, xres = rnd(Mh, 1024), (torch.randn(256, 1024, device=dev) / 32).to(dt), rnd(Mh, 256)
a3, wl = rnd(Mh, 4864), (torch.randn(256, 4864, device=dev) / 70).to(dt)
cases = [
    ("conv2 fwd 134k x 256 x 2304", lambda: ops.conv2_fwd(y1, w2p, bias=b2, act=ops.ACT_RELU)),
    ("conv2 dgrad", lambda: ops.conv2_dgrad_kc(dy2, wt, y1)),
    ("ctc head + lse 35145 x 10000 x 256", lambda: ops.gemm_nt_lse(eo, wh, bh)),
    ("rnnt head fwd 281600 x 1000 x 512", lambda: ops.rnnt_head_forward(h, wo, bo, B, T, U, labels, elens, ylens, 0)),
    ("rnnt head grad chunk 65536 x 1000 x 512", lambda: ops.rnnt_head_grad(h[:CH], wo, bo, coef[:CH], ycol[:CH], 0, dzp[:, :V])),
    ("joint dgrad 65536 x 512 x 1024", lambda: ops.gemm_nt(dzp, w_t, out=dpre, dact_pre=h[:CH], dact=ops.DACT_TANH_OUT)),
    ("ffn2 35145 x 256 x 1024 (residual)", lambda: ops.gemm_nt(a2, wf2, out=xres, residual=xres, res_scale=0.5, drop_p=0.1, seed=3)),
    ("lin 35145 x 256 x 4864", lambda: ops.gemm_nt(a3, wl)),
]
for name, fn in cases:
    row = f"{name:42s}"
    for waves in ((8, 4) if os.environ.get("WAVES4") else (8,)):
        for bm in (0, 128, 192, 256):
            lib.set_option("big_waves", waves)
            lib.set_option("big_bm", bm)
            row += f" | w{waves} bm{bm:<3d} {graph_time(fn, n=5):7.1f}"
    print(row, flush=True)
lib.set_option("big_waves", 8)
lib.set_option("big_bm", 0)
