"""NT / NN GEMMs of the encoder shapes under each block-tile override (2 = 128x64, 3 = 64x64, 0 = automatic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emoasr_amd import lib, ops
dev = torch.device("cuda:0")
dt = torch.bfloat16
M = int(os.environ.get("M", 6840))


def timeit(f, n=40):
    for _ in range(8):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for (N, K) in ((1024, 256), (256, 1024), (768, 256), (512, 256), (256, 256), (256, 512)):
    a = torch.randn(M, K, device=dev).to(dt)
    bnt = torch.randn(N, K, device=dev).to(dt)
    bnn = torch.randn(K, N, device=dev).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    bias = torch.randn(N, device=dev)
    row = []
    for tile in (0, 2, 3):
        lib.set_option("gemm_tile", tile)
        row.append((tile, timeit(lambda: ops.gemm_nt(a, bnt, out=out, bias=bias)), timeit(lambda: ops.gemm_nn(a, bnn, out=out))))
    lib.set_option("gemm_tile", 0)
    print(f"M={M} N={N} K={K}: " + "  ".join(f"tile{t}: nt {x:5.1f} nn {y:5.1f}" for t, x, y in row), flush=True)
