import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emoasr_amd import ops
from tools._timing import graph_time
dev = torch.device("cuda:0")
bf = torch.bfloat16
for (M, N, K) in [(10, 768, 256), (10, 256, 256), (10, 1024, 256), (10, 256, 1024), (10, 10000, 256)]:
    x = torch.randn(M, K, device=dev).to(bf)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf)
    b = torch.randn(N, device=dev)
    g, be = torch.ones(K, device=dev), torch.zeros(K, device=dev)
    gr, br = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    res = torch.randn(M, N, device=dev).to(bf)
    t0 = graph_time(lambda: ops.rowlin(x, w, b), n=10)
    t1 = graph_time(lambda: ops.rowlin(x, w, b, ln_a=(g, be)), n=10)
    t2 = graph_time(lambda: ops.rowlin(x, w, b, res=res, ln_r=(gr, br)), n=10) if N <= 1024 else 0.0
    t3 = graph_time(lambda: ops.gemm_nt(x, w, bias=b), n=10)
    print(f"M {M} N {N} K {K}: rowlin {t0:.1f} us, +LN prologue {t1:.1f}, +LN residual {t2:.1f}; gemm_nt 64x64 {t3:.1f}", flush=True)
