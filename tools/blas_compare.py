"""Reference point only (not on the product path): what the vendor BLAS behind torch.matmul does on the L2 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emoasr_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 7200
for name, m, n, k in [("ffn1", M, 1024, 256), ("ffn2", M, 256, 1024), ("qkv", M, 768, 256), ("out", M, 256, 256), ("head", M, 10000, 256),
                      ("dcol", 136800, 2304, 256)]:
    a = torch.randn(m, k, device=dev).bfloat16(); b = torch.randn(n, k, device=dev).bfloat16()
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    t_mine = timeit(lambda: ops.gemm_nt(a, b, out=out))
    bt = b.t()
    t_blas = timeit(lambda: torch.mm(a, bt, out=out))
    print(f"{name:6s} {m}x{n}x{k}: mine {t_mine:7.1f} us   torch.mm {t_blas:7.1f} us")
x = torch.randn(M, 256, device=dev).bfloat16(); y = torch.empty_like(x)
print(f"empty-ish kernel (scale_dropout p=0 on 7200x256): {timeit(lambda: ops.scale_dropout(x, 1.0, 0.0, 0)):.1f} us")
