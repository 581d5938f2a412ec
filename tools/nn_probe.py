"""time NT vs NN (dgrad) GEMMs of the FFN shapes with and without the fused activation-derivative epilogue"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emoasr_amd import ops
dev = torch.device("cuda:0")
dt = torch.bfloat16
M = int(os.environ.get("M", 6840))


def timeit(f, n=50):
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for (N, K) in ((1024, 256), (256, 1024), (768, 256), (256, 256)):
    a = torch.randn(M, K, device=dev).to(dt)
    bnt = torch.randn(N, K, device=dev).to(dt)
    bnn = torch.randn(K, N, device=dev).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    pre = torch.randn(M, N, device=dev).to(dt)
    bias = torch.randn(N, device=dev)
    fl = 2.0 * M * N * K
    rows = [("nt plain", lambda: ops.gemm_nt(a, bnt, out=out)),
            ("nt bias+swish", lambda: ops.gemm_nt(a, bnt, out=out, bias=bias, act=ops.ACT_SWISH)),
            ("nn plain", lambda: ops.gemm_nn(a, bnn, out=out)),
            ("nn dact swish", lambda: ops.gemm_nn(a, bnn, out=out, dact_pre=pre, dact=ops.ACT_SWISH)),
            ("nn dact+drop", lambda: ops.gemm_nn(a, bnn, out=out, dact_pre=pre, dact=ops.ACT_SWISH, drop_p=0.1, seed=123))]
    for name, f in rows:
        us = timeit(f)
        print(f"M={M} N={N} K={K} {name:16s} {us:7.1f} us  {fl / us * 1e-6:6.1f} TF/s", flush=True)
