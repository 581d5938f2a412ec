"""Time the attention entry points alone at the L2 shape (B=24, T'=299)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import math, torch
from emoasr_amd import ops
dev = torch.device("cuda:0")
B, T, H, D = int(os.environ.get("B", 24)), int(os.environ.get("T", 299)), 4, 256
dt = torch.bfloat16
qkv = torch.randn(B, T, 3 * D, device=dev).to(dt)
q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
pos = torch.randn(2 * T - 1, D, device=dev).to(dt)
bu, bv = torch.randn(D, device=dev) * 0.1, torch.randn(D, device=dev) * 0.1
klens = torch.full((B,), T, device=dev, dtype=torch.int32)
scale = 1 / math.sqrt(64)
p = float(os.environ.get("P", 0.1))
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out, lse = ops.attn_fwd(q, k, v, H, scale, pos=pos, bias_u=bu, bias_v=bv, klens=klens, drop_p=p, seed=1)
print(f"fwd {timeit(lambda: ops.attn_fwd(q, k, v, H, scale, pos=pos, bias_u=bu, bias_v=bv, klens=klens, drop_p=p, seed=1)):.1f} us")
dout = torch.randn(B, T, D, device=dev).to(dt)
dqkv = torch.empty_like(qkv)
dq, dk, dv = dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:]
dpos = torch.zeros(2 * T - 1, D, device=dev); dbu = torch.zeros(D, device=dev); dbv = torch.zeros(D, device=dev)
sc = ops.AttnScratch(B, H, T, T, dt, dev, True)
for mat in (True, False):
    f = lambda: ops.attn_bwd(dout, out, lse, q, k, v, H, scale, dq, dk, dv, pos=pos, bias_u=bu, bias_v=bv, klens=klens,
                             drop_p=p, seed=1, dpos=dpos, dbias_u=dbu, dbias_v=dbv, scratch=sc if mat else None, materialise=mat)
    print(f"bwd materialise={mat}: {timeit(f):.1f} us")
