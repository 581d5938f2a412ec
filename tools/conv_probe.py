"""The Conv2d(256->256, k3, s2) products of one micro-batch (16 utterances x ~1750 frames): time per launch; run under
`rocprofv3 --pmc FETCH_SIZE` (or WRITE_SIZE) and list with tools/pmc_list.py for the bytes each one pulls through L2.
Operand sizes are printed so the counter can be compared with the algorithmic bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emoasr_amd import lib, ops

dev = torch.device("cuda:0")
dt = torch.bfloat16
C = 256
B, T = int(os.environ.get("B", 16)), int(os.environ.get("T", 1757))
REP = int(os.environ.get("REP", 5))
torch.manual_seed(0)
x = torch.randn(B, T, 80, device=dev)
w1, b1 = torch.randn(C, 9, device=dev) * 0.3, torch.randn(C, device=dev) * 0.1
w2 = (torch.randn(C, C, 3, 3, device=dev) * (9 * C) ** -0.5).to(dt)
b2 = torch.randn(C, device=dev) * 0.1
y1 = ops.conv1_fwd(x, w1, b1, dt)
w2p = w2.permute(0, 2, 3, 1).reshape(C, 9 * C).contiguous()
wt = w2.permute(1, 2, 3, 0).reshape(C, 9 * C).contiguous()
y2 = ops.conv2_fwd(y1, w2p, bias=b2, act=ops.ACT_RELU)
dy2 = torch.randn_like(y2)
dw = torch.zeros(C, 9 * C, device=dev)
db = torch.zeros(C, device=dev)
print(f"y1 {tuple(y1.shape)} {y1.numel() * 2 / 1e6:.0f} MB, y2 / dy2 {tuple(y2.shape)} {y2.numel() * 2 / 1e6:.0f} MB, weight {w2p.numel() * 2 / 1e6:.1f} MB")

def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REP * 1e3

for name, opts in [("default", {}), ("tap-major k order", {"big_korder": 0})] + [(f"big_bm {b}", {"big_bm": b}) for b in (128, 192, 256)]:
    for k, v in opts.items(): lib.set_option(k, v)
    u1 = timeit(lambda: ops.conv2_fwd(y1, w2p, bias=b2, act=ops.ACT_RELU))
    u2 = timeit(lambda: ops.conv2_dgrad_kc(dy2, wt, y1))
    u3 = timeit(lambda: ops.conv2_wgrad(dy2, y1, dw, db, accumulate=True))
    print(f"{name:20s} fwd {u1:7.1f} us   dgrad {u2:7.1f} us   wgrad {u3:7.1f} us   ({REP + 1} launches each)")
    for k in opts: lib.set_option(k, {"big_korder": 1}.get(k, 0))
