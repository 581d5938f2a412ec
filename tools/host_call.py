"""Host cost of one op call (launch-return, GPU kept far behind by the queue depth being irrelevant: we time
the CPU side only, over many calls, and sync at the end)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctypes import byref, c_void_p
from emoasr_amd import ops, lib
dev = torch.device("cuda:0")
M = 512
a = torch.randn(M, 256, device=dev).bfloat16(); b = torch.randn(256, 256, device=dev).bfloat16()
bias = torch.randn(256, device=dev); out = torch.empty(M, 256, device=dev, dtype=torch.bfloat16)
g = torch.ones(256, device=dev)
def bench(name, fn, n=3000):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{name:44s} {1e6 * dt / n:6.2f} us/call")
with ops.stream_scope():
    bench("ops.gemm_nt(out=None, bias, residual)", lambda: ops.gemm_nt(a, b, bias=bias, residual=a, res_scale=1.0))
    bench("ops.gemm_nt(out=out)", lambda: ops.gemm_nt(a, b, out=out))
    bench("ops.layernorm_fwd", lambda: ops.layernorm_fwd(a, g, g, 1e-5, True))
    bench("torch.empty", lambda: torch.empty(M, 256, device=dev, dtype=torch.bfloat16))
    bench("torch.empty_like", lambda: torch.empty_like(a))
    bench("make_epilogue", lambda: ops.make_epilogue(bias=bias, residual=a))
    ep = ops.make_epilogue(bias=bias)
    st = ops._stream()
    args = (lib.BF16, M, 256, 256, ops._p(a), 256, ops._p(b), 256, ops._p(out), 256, byref(ep), st)
    bench("lib.call prebuilt args", lambda: lib.call("emoasr_gemm_nt", *args))
    fn = lib.load().emoasr_gemm_nt
    bench("raw ctypes fn prebuilt args", lambda: fn(*args))
    iargs = (lib.BF16, M, 256, 256, a.data_ptr(), 256, b.data_ptr(), 256, out.data_ptr(), 256, byref(ep), st)
    bench("raw ctypes fn int pointers", lambda: fn(*iargs))
    bench("6x _p()", lambda: (ops._p(a), ops._p(b), ops._p(out), ops._p(bias), ops._p(a), ops._p(g)))
    bench("6x data_ptr()", lambda: (a.data_ptr(), b.data_ptr(), out.data_ptr(), bias.data_ptr(), a.data_ptr(), g.data_ptr()))
    bench("_rows x3", lambda: (ops._rows(a), ops._rows(b), ops._rows(out)))
