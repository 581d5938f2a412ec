import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from emoasr_amd import ops, lib
dev = torch.device("cuda:0")
M, N, K = [int(v) for v in os.environ.get("SHAPE", "136800,2304,256").split(",")]
a = torch.randn(M, K, device=dev).bfloat16(); b = torch.randn(K, N, device=dev).bfloat16(); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(3): ops.gemm_nn(a, b, out=out)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (4096 * 8))()
L = lib.load(); L.emoasr_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.emoasr_debug_trace(buf, 4096 * 8)
t = np.array(buf[:], dtype=np.float64).reshape(4096, 8)[:, :6]
t = t[t[:, 0] > 0]
d = np.diff(t, axis=1)
print("blocks", len(t), "clock units (s_memtime ticks = 100 MHz? or shader cycles): median per phase")
print("  [0->1 setup, 1->2 first loads+store+sync, 2->3 k loop, 3->4 epilogue issue, 4->5 drain stores]")
print("  median", np.median(d, axis=0), " mean", d.mean(axis=0).round(0))
print("  total per block median", np.median(t[:, 5] - t[:, 0]))
order = np.argsort(t[:, 0]); print("  span of first 4096 block starts", t[order[-1], 0] - t[order[0], 0])
