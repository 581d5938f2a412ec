"""Device time of the Conv2d(256->256,k3,s2) front-end products at a 27 k-frame batch (M = B*T2*F2 ~ 133 k): the large-tile
kernel (csrc/gemm_big.hip) against the 128x64-tile implicit GEMMs it replaces, and the plain 136800x2304x256 product
against the vendor BLAS (torch.mm).  HIP-graph timed."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from emoasr_amd import lib, ops
from tools._timing import graph_time

dev = torch.device("cuda:0")
dt = torch.bfloat16
C = 256
B, T = int(os.environ.get("B", 23)), int(os.environ.get("T", 1200))
torch.manual_seed(0)
x = torch.randn(B, T, 80, device=dev)
w1, b1 = torch.randn(C, 9, device=dev) * 0.3, torch.randn(C, device=dev) * 0.1
w2 = (torch.randn(C, C, 3, 3, device=dev) * (9 * C) ** -0.5).to(dt)
b2 = torch.randn(C, device=dev) * 0.1
y1 = ops.conv1_fwd(x, w1, b1, dt)
w2p = w2.permute(0, 2, 3, 1).reshape(C, 9 * C).contiguous()
wt = w2.permute(1, 2, 3, 0).reshape(C, 9 * C).contiguous()
y2 = ops.conv2_fwd(y1, w2p, bias=b2, act=ops.ACT_RELU)
M = y2.numel() // C
flop = 2.0 * M * 9 * C * C
dy2 = torch.randn_like(y2)
print(f"B {B} T {T}: y1 {tuple(y1.shape)} M = {M} rows, {flop / 1e9:.1f} GFLOP per product", flush=True)
bms = [int(v) for v in os.environ.get("BMS", "0,256,192,128").split(",")]
for bm in bms:
    lib.set_option("big_bm", bm)
    us = graph_time(lambda: ops.conv2_fwd(y1, w2p, bias=b2, act=ops.ACT_RELU), n=5)
    print(f"conv2 fwd   big bm={bm:3d}: {us:7.1f} us  {flop / us / 1e6:6.0f} TF/s", flush=True)
    us = graph_time(lambda: ops.conv2_dgrad_kc(dy2, wt, y1), n=5)
    print(f"conv2 dgrad big bm={bm:3d}: {us:7.1f} us  {flop / us / 1e6:6.0f} TF/s", flush=True)
lib.set_option("big_bm", 0)
lib.set_option("conv_big", 0)
us = graph_time(lambda: ops.conv2_fwd(y1, w2p, bias=b2, act=ops.ACT_RELU), n=5)
print(f"conv2 fwd   128x64     : {us:7.1f} us  {flop / us / 1e6:6.0f} TF/s", flush=True)
lib.set_option("conv_big", 1)
us = graph_time(lambda: ops.conv2_dgrad(dy2, w2p, y1), n=5)
print(f"conv2 dgrad 128x64 x4  : {us:7.1f} us  {flop / us / 1e6:6.0f} TF/s", flush=True)
dw = torch.zeros(C, 9 * C, device=dev)
us = graph_time(lambda: ops.conv2_wgrad(dy2, y1, dw, accumulate=True), n=5)
print(f"conv2 wgrad tn         : {us:7.1f} us  {flop / us / 1e6:6.0f} TF/s", flush=True)
# plain product of the same size
for (m, n, k) in [(136800, 256, 2304), (7029, 1024, 256), (7029, 768, 256), (7029, 512, 256), (5000, 1024, 256), (9000, 512, 256)]:
    a = torch.randn(m, k, device=dev).to(dt)
    b = (torch.randn(n, k, device=dev) * k ** -0.5).to(dt)
    f = 2.0 * m * n * k
    out = torch.empty(m, n, device=dev, dtype=dt)
    for bm in bms:
        lib.set_option("big_bm", bm)
        us = graph_time(lambda: ops.gemm_nt_big(a, b, out=out), n=5)
        print(f"gemm {m}x{n}x{k} big bm={bm:3d}: {us:7.1f} us {f / us / 1e6:6.0f} TF/s", flush=True)
    lib.set_option("big_bm", 0)
    us = graph_time(lambda: ops.gemm_nt(a, b, out=out), n=5)
    print(f"gemm {m}x{n}x{k} 128x64      : {us:7.1f} us {f / us / 1e6:6.0f} TF/s", flush=True)
    bt = b.t().contiguous()
    us = graph_time(lambda: torch.mm(a, bt, out=out), n=5)
    print(f"gemm {m}x{n}x{k} blas (NN)   : {us:7.1f} us {f / us / 1e6:6.0f} TF/s", flush=True)
    us = graph_time(lambda: torch.mm(a, b.t(), out=out), n=5)
    print(f"gemm {m}x{n}x{k} blas (NT)   : {us:7.1f} us {f / us / 1e6:6.0f} TF/s", flush=True)
