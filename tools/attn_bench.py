"""Device time of the attention entry points alone at L2 training-batch shapes (HIP-graph timed: 10 calls per replay).
Shapes: (B, T') pairs the bench's sampler produces -- ragged key lengths drawn like a sorted-bucket batch."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from emoasr_amd import ops
from tools._timing import graph_time

dev = torch.device("cuda:0")
H, D = 4, 256
DT = os.environ.get("DT", "bf16")   # bf16 | f32 | f32x3 (f32 storage, split-bf16 products)
dt = torch.bfloat16 if DT == "bf16" else torch.float32
ops.split_products(bool(1 if DT == "f32x3" else 0))
p = float(os.environ.get("P", 0.1))
shapes = [(22, 320), (24, 299), (12, 590), (8, 875), (40, 170), (110, 320), (60, 590), (200, 170)]   # the last three: five stacked micro-batches' worth
if os.environ.get("B"):
    shapes = [(int(os.environ["B"]), int(os.environ.get("T", 299)))]
modes = os.environ.get("MODES", "fused,fused1,mat" if DT == "bf16" else "mat").split(",")
scale = 1 / math.sqrt(64)
for B, T in shapes:
    torch.manual_seed(0)
    qkv = torch.randn(B, T, 3 * D, device=dev).to(dt)
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    pos = torch.randn(2 * T - 1, D, device=dev).to(dt)
    bu, bv = torch.randn(D, device=dev) * 0.1, torch.randn(D, device=dev) * 0.1
    lens = sorted((int(T * (0.88 + 0.12 * i / max(B - 1, 1))) for i in range(B)), reverse=True)
    klens = torch.tensor(lens, device=dev, dtype=torch.int32)
    pairs = sum(l * T for l in lens) * H
    # the training path's keep mask as bits (hashed once per layer, up front: not part of the timed calls); MASK=0: inline hash
    km = ops.attn_dropmask(q, k, H, klens=klens, drop_p=p, seed=1) if (p > 0 and DT == "bf16" and os.environ.get("MASK", "1") != "0") else None
    out, lse = ops.attn_fwd(q, k, v, H, scale, pos=pos, bias_u=bu, bias_v=bv, klens=klens, drop_p=p, seed=1, keep_mask=km)
    fw = graph_time(lambda: ops.attn_fwd(q, k, v, H, scale, pos=pos, bias_u=bu, bias_v=bv, klens=klens, drop_p=p, seed=1, keep_mask=km), n=10)
    dout = torch.randn(B, T, D, device=dev).to(dt)
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:]
    dpos = torch.zeros(2 * T - 1, D, device=dev)
    dbu, dbv = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    sc = ops.AttnScratch(B, H, T, T, dt, dev, True)
    row = f"B {B:3d} T' {T:4d}: fwd {fw:7.1f} us ({3 * 2 * 64 * pairs / fw / 1e6:5.0f} TF/s)"
    for mode in modes:   # fused: the two-pass backward (default); fused1: the single-pass kernel of rounds 2-3; mat: materialised
        from emoasr_amd import lib
        lib.set_option("attn_bwd_split", 0 if mode == "fused1" else 1)
        mat = "fused" if mode in ("fused", "fused1") else (mode == "mat")
        f = lambda: ops.attn_bwd(dout, out, lse, q, k, v, H, scale, dq, dk, dv, pos=pos, bias_u=bu, bias_v=bv, klens=klens,
                                 drop_p=p, seed=1, dpos=dpos, dbias_u=dbu, dbias_v=dbv, scratch=sc if mat is True else None,
                                 materialise=mat, keep_mask=km if mode == "fused" else None)
        us = graph_time(f, n=10)
        # 9 matmul units of 2*64 flop per valid (query, key, head) pair: S, band x2, dP, dV, dK, dQu, dQv x2
        row += f"   bwd {mode}: {us:7.1f} us ({9 * 2 * 64 * pairs / us / 1e6:5.0f} TF/s)"
    print(row, flush=True)
