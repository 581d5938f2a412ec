"""Micro-benchmark of the GEMM entry points on the L2 model's shapes (B*T' = 7200 rows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emoasr_amd import ops

dev = torch.device("cuda:0")
from emoasr_amd import lib as _lib
for _o in ('gemm_tile', 'gemm_kb', 'gemm_xcd', 'big_min_tiles'):
    if os.environ.get(_o.upper()) is not None: _lib.set_option(_o, int(os.environ[_o.upper()]))
M = int(os.environ.get("M", 7200))
dt = torch.bfloat16
shapes_nt = [("ffn1", M, 1024, 256), ("ffn2", M, 256, 1024), ("qkv", M, 768, 256), ("out", M, 256, 256),
             ("pw1", M, 512, 256), ("head", M, 10000, 256), ("lin", M, 256, 4864)]
if os.environ.get("NO_DCOL"):
    pass
shapes_nn = [("d_ffn2", M, 1024, 256), ("d_ffn1", M, 256, 1024), ("d_qkv", M, 256, 768), ("d_out", M, 256, 256),
             ("d_pw1", M, 256, 512), ("d_head", M, 256, 10000), ("d_lin", M, 4864, 256)]
shapes_tn = [("w_ffn1", 1024, 256, M), ("w_ffn2", 256, 1024, M), ("w_qkv", 768, 256, M), ("w_out", 256, 256, M),
             ("w_head", 10000, 256, M), ("w_lin", 256, 4864, M)]

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us

def rnd(*s): return torch.randn(*s, device=dev).to(dt)
tot = 0.0
for name, m, n, k in shapes_nt:
    a, b = rnd(m, k), rnd(n, k); out = torch.empty(m, n, device=dev, dtype=dt); bias = torch.randn(n, device=dev)
    us = timeit(lambda: ops.gemm_nt(a, b, out=out, bias=bias))
    if os.environ.get("EPI") and k == 256:   # the training epilogues: w1 (bias, Swish, dropout, saved pre-activation) / residual
        pre = torch.empty_like(out); resid = rnd(m, n)
        u1 = timeit(lambda: ops.gemm_nt(a, b, out=out, bias=bias, act=ops.ACT_SWISH, pre_out=pre, drop_p=0.1, seed=5))
        u2 = timeit(lambda: ops.gemm_nt(a, b, out=out, bias=bias, residual=resid, res_scale=0.5, drop_p=0.1, seed=5))
        u4 = timeit(lambda: ops.gemm_nt(a, b, out=out, bias=bias, act=ops.ACT_SWISH | ops.ACT_SAVE_DACT, pre_out=pre, drop_p=0.1, seed=5))
        print(f"   {name:8s} with swish+dropout+pre_out {u1:8.1f} us   saved factor instead of pre {u4:8.1f} us   with residual+dropout {u2:8.1f} us")
    if os.environ.get("EPI") and n == 256:    # the block outputs: x += 0.5 * dropout(acc + bias), in place (C aliases the residual)
        x = rnd(m, n)
        u3 = timeit(lambda: ops.gemm_nt(a, b, out=x, bias=bias, residual=x, res_scale=0.5, drop_p=0.1, seed=5))
        print(f"   {name:8s} in-place residual + dropout {u3:8.1f} us")
    tot += us
    ub = timeit(lambda: torch.mm(a, b.t(), out=out)) if os.environ.get("BLAS") else 0.0   # reference point only (hipBLASLt)
    print(f"nt {name:8s} {m}x{n}x{k}: {us:8.1f} us  {2*m*n*k/us/1e6:7.1f} TF/s  {(m*k+n*k+m*n)*2/us/1e3:7.1f} GB/s   blas {ub:8.1f} us")
for name, m, n, k in shapes_nn:
    a, b = rnd(m, k), rnd(k, n); out = torch.empty(m, n, device=dev, dtype=dt)
    us = timeit(lambda: ops.gemm_nn(a, b, out=out))
    if os.environ.get("EPI") and k == 256 and n == 1024:   # dgrad through w2 with the Swish derivative of the saved pre-activation
        pre = rnd(m, n)
        u3 = timeit(lambda: ops.gemm_nn(a, b, out=out, dact_pre=pre, dact=ops.ACT_SWISH, drop_p=0.1, seed=5))
        u5 = timeit(lambda: ops.gemm_nn(a, b, out=out, dact_pre=pre, dact=ops.DACT_MUL))
        print(f"   {name:8s} with dact(pre) + dropout {u3:8.1f} us   with the saved factor {u5:8.1f} us")
    tot += us
    ub = timeit(lambda: torch.mm(a, b, out=out)) if os.environ.get("BLAS") else 0.0
    print(f"nn {name:8s} {m}x{n}x{k}: {us:8.1f} us  {2*m*n*k/us/1e6:7.1f} TF/s  {(m*k+n*k+m*n)*2/us/1e3:7.1f} GB/s   blas {ub:8.1f} us")
for name, n1, n2, k in shapes_tn:
    a, b = rnd(k, n1), rnd(k, n2); out = torch.zeros(n1, n2, device=dev); cs = torch.zeros(n1, device=dev)
    us = timeit(lambda: ops.gemm_tn(a, b, out=out, accumulate=True, colsum=cs))
    tot += us
    ob = torch.zeros(n1, n2, device=dev, dtype=dt)
    ub = timeit(lambda: torch.mm(a.t(), b, out=ob)) if os.environ.get("BLAS") else 0.0
    print(f"tn {name:8s} {n1}x{n2}x{k}: {us:8.1f} us  {2*n1*n2*k/us/1e6:7.1f} TF/s   blas {ub:8.1f} us")
print(f"sum {tot:.1f} us")
