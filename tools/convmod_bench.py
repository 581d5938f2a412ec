"""Device time of the convolution-module element-wise / stencil kernels at an L2 batch shape (HIP-graph timed):
the fused kernels of csrc/convfused.hip against the launches they replace."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from emoasr_amd import lib, ops
from tools._timing import graph_time

dev = torch.device("cuda:0")
B, T, C, K = int(os.environ.get("B", 22)), int(os.environ.get("T", 320)), 256, 31
dt_ = torch.bfloat16
g = torch.randn(B * T, 2 * C, device=dev).to(dt_)
w, bias = torch.randn(C, K, device=dev) * K ** -0.5, torch.randn(C, device=dev) * 0.1
gamma, beta = 1 + 0.1 * torch.randn(C, device=dev), 0.1 * torch.randn(C, device=dev)
ds = torch.randn(B * T, C, device=dev).to(dt_)
rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
dgam, dbet = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
dw, db = torch.zeros(C, K, device=dev), torch.zeros(C, device=dev)
gl = ops.glu_fwd(g)
c, mean, var = ops.dwconv_bn_stats_fwd(gl.view(B, T, C), w, bias, rm, rv, 0.1)
c2 = c.view(B * T, C)
dc = ops.bn_swish_bwd(ds, c2, mean, var, gamma, beta, 1e-5, dgam, dbet)
dgl = ops.dwconv_bwd_x(dc.view(B, T, C), w)
rows = []
for lds in (0, 1):
    lib.set_option("dwconv_lds", lds)
    tag = "lds" if lds else "r1 "
    rows.append((f"glu_fwd", graph_time(lambda: ops.glu_fwd(g), n=10)))
    rows.append((f"dwconv+stats+finalize [{tag}]", graph_time(lambda: ops.dwconv_bn_stats_fwd(gl.view(B, T, C), w, bias, rm, rv, 0.1), n=10)))
    rows.append((f"dwconv_bwd_x [{tag}]", graph_time(lambda: ops.dwconv_bwd_x(dc.view(B, T, C), w), n=10)))
lib.set_option("dwconv_lds", 1)
rows.append(("bn_swish_bwd (sums+fold+apply)", graph_time(lambda: ops.bn_swish_bwd(ds, c2, mean, var, gamma, beta, 1e-5, dgam, dbet), n=10)))
rows.append(("dwconv_bwd_w (+reduce)", graph_time(lambda: ops.dwconv_bwd_w(dc.view(B, T, C), gl.view(B, T, C), dw, db, accumulate=True), n=10)))
rows.append(("glu_bwd", graph_time(lambda: ops.glu_bwd(g, dgl.view(B * T, C)), n=10)))
rows.append(("FUSED glu_dwconv_fwd (+finalize)", graph_time(lambda: ops.glu_dwconv_fwd(g, B, T, w, bias, rm, rv, 0.1, None, True), n=10)))
rows.append(("FUSED conv_bwd (sums+fold+fused+reduce)", graph_time(lambda: ops.conv_bwd_fused(ds, c2, mean, var, gamma, beta, 1e-5, dgam, dbet, g, w, dw, db, B, T), n=10)))
for k, v in rows:
    print(f"{k:45s} {v:7.1f} us")
