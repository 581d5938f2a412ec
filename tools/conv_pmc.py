"""A few plain (non-graph) launches of the large-tile conv2 forward / data gradient for rocprofv3 --pmc passes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from emoasr_amd import lib, ops

dev = torch.device("cuda:0")
dt = torch.bfloat16
C = 256
B, T = int(os.environ.get("B", 23)), int(os.environ.get("T", 1200))
torch.manual_seed(0)
x = torch.randn(B, T, 80, device=dev)
w1, b1 = torch.randn(C, 9, device=dev) * 0.3, torch.randn(C, device=dev) * 0.1
w2 = (torch.randn(C, C, 3, 3, device=dev) * (9 * C) ** -0.5).to(dt)
b2 = torch.randn(C, device=dev) * 0.1
y1 = ops.conv1_fwd(x, w1, b1, dt)
w2p = w2.permute(0, 2, 3, 1).reshape(C, 9 * C).contiguous()
wt = w2.permute(1, 2, 3, 0).reshape(C, 9 * C).contiguous()
lib.set_option("big_korder", int(os.environ.get("KORDER", 1)))
for _ in range(3):
    y2 = ops.conv2_fwd(y1, w2p, bias=b2, act=ops.ACT_RELU)
    dy1 = ops.conv2_dgrad_kc(y2, wt, y1)
torch.cuda.synchronize()
