import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emoasr_amd import ops
dev = torch.device("cuda:0")
M, N, K = [int(v) for v in os.environ.get("SHAPE", "7200,1024,256").split(",")]
kind = os.environ.get("KIND", "nt")
dt = torch.bfloat16
a = torch.randn(M, K, device=dev).to(dt)
if kind == "nt":
    b = torch.randn(N, K, device=dev).to(dt); out = torch.empty(M, N, device=dev, dtype=dt); bias = torch.randn(N, device=dev)
    f = lambda: ops.gemm_nt(a, b, out=out, bias=bias)
elif kind == "nn":
    b = torch.randn(K, N, device=dev).to(dt); out = torch.empty(M, N, device=dev, dtype=dt)
    f = lambda: ops.gemm_nn(a, b, out=out)
else:
    a = torch.randn(K, M, device=dev).to(dt); b = torch.randn(K, N, device=dev).to(dt); out = torch.zeros(M, N, device=dev)
    f = lambda: ops.gemm_tn(a, b, out=out, accumulate=True)
for _ in range(20): f()
torch.cuda.synchronize()
