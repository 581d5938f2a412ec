"""Do the long-K, N = 256 products of the stacked step (ffn2 / d_ffn1 35145 x 256 x 1024, d_qkv K = 768, the front-end Linear
K = 4864, d_head K = 10000) run faster on the large-tile kernel (one 256-column tile per row band: every A row read once) than on
the 64 x 64 kernel?  HIP-graph timed; torch.mm (hipBLASLt) as the reference point."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from emoasr_amd import lib, ops
from tools._timing import graph_time

dev = torch.device("cuda:0")
M = int(os.environ.get("M", 35145))
dt = torch.bfloat16
rnd = lambda *s: torch.randn(*s, device=dev).to(dt)
for name, N, K in (("ffn2/d_ffn1", 256, 1024), ("d_qkv", 256, 768), ("d_pw1", 256, 512), ("lin", 256, 4864), ("d_head", 256, 10048)):
    a, b, bias = rnd(M, K), rnd(N, K), torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=dt)
    t0 = graph_time(lambda: ops.gemm_nt(a, b, out=out, bias=bias))
    ref = out.clone()
    row = f"{name:12s} {M}x{N}x{K}: 64x64 {t0:7.1f} us"
    for bm in (128, 192, 256):
        lib.set_option("big_bm", bm)
        t = graph_time(lambda: ops.gemm_nt_big(a, b, out=out, bias=bias))
        err = (out.float() - ref.float()).abs().max().item()
        row += f" | big bm={bm} {t:7.1f} us (diff {err:.1e})"
    lib.set_option("big_bm", 0)
    tb = graph_time(lambda: torch.mm(a, b.t(), out=out))
    print(row + f" | blas {tb:7.1f} us   [{2.0 * M * N * K / 1e6:.0f} MFLOP]")
