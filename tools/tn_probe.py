"""Weight-gradient (TN) products of one encoder layer at the stacked row count, one grouped launch, under a few launch policies.
Prints the time per launch; run under `rocprofv3 --pmc FETCH_SIZE` to get the bytes each variant pulls through L2 (dispatch order =
the order printed here).  usage: python tools/tn_probe.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emoasr_amd import ops, lib

dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 35145
REP = int(os.environ.get("REP", 5))
dt = torch.bfloat16
def rnd(*s): return torch.randn(*s, device=dev).to(dt)
# the nine products of csrc/layer.hip's backward, in its order: (N1, N2, rows, bias gradient)
R = int(os.environ.get("POS_ROWS", 4400))
shapes = [(256, 1024, K, 1), (1024, 256, K, 1), (256, 256, K, 1), (512, 256, K, 1), (256, 256, K, 1), (256, 256, R, 0), (768, 256, K, 1),
          (256, 1024, K, 1), (1024, 256, K, 1)]
if os.environ.get("UNIFORM"):
    shapes = [(n1, n2, K, 1) for n1, n2, _, _ in shapes]
probs = []
for n1, n2, k, cs in shapes:
    probs.append((rnd(k, n1), rnd(k, n2), torch.zeros(n1, n2, device=dev), 1.0, torch.zeros(n1, device=dev) if cs and not os.environ.get("NO_COLSUM") else None, 1.0))
alg = sum(k * (n1 + n2) * 2 for n1, n2, k, _ in shapes)
print(f"rows {K}: algorithmic input bytes per grouped launch {alg/1e6:.1f} MB")

def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REP * 1e3

variants = [("default", {}), ("slices placed per XCD", {"tn_place": 1}), ("hw block order", {"gemm_xcd": 0}),
            ("BK=64, 512 blocks", {"gemm_kb": 2, "tn_group_blocks": 512}), ("BK=64, 256 blocks", {"gemm_kb": 2, "tn_group_blocks": 256}),
            ("BK=32, 512 blocks", {"tn_group_blocks": 512}), ("BK=32, 896 blocks", {"tn_group_blocks": 896}), ("64x64 tiles", {"gemm_tile": 3})]
for name, opts in variants:
    for k, v in opts.items(): lib.set_option(k, v)
    us = timeit(lambda: ops.gemm_tn_grouped(probs))
    print(f"group  {name:16s} {us:8.1f} us  {alg/us/1e3:7.1f} GB/s algorithmic   ({REP + 1} launches)")
    for k in opts: lib.set_option(k, {"gemm_xcd": 1}.get(k, 0))
for i in (0, 1, 2, 3):
    us = timeit(lambda: ops.gemm_tn_grouped(probs[i:i + 1]))
    n1, n2 = shapes[i][:2]
    print(f"single {n1}x{n2}        {us:8.1f} us  {K*(n1+n2)*2/us/1e3:7.1f} GB/s algorithmic   ({REP + 1} launches)")

# the two plain launches of the step: CTC head weight gradient (N1 = vocab) and the Conv2d weight gradient (implicit GEMM, K = B*T2*F2)
a, b = rnd(K, 10000), rnd(K, 256)
out, cs = torch.zeros(10000, 256, device=dev), torch.zeros(10000, device=dev)
B_, T1, F1, C = 16, 4 * (K // 5 // 16) // 2, 39, 256
y1 = rnd(B_, T1, F1, C); T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
dy2 = rnd(B_, T2, F2, C); dw = torch.zeros(C, 3, 3, C, device=dev); db = torch.zeros(C, device=dev)
print(f"conv2 wgrad: K = {B_ * T2 * F2}")
for name, opts in [("default", {}), ("BK=64, 512", {"gemm_kb": 2, "tn_group_blocks": 512}), ("288 blocks", {"tn_group_blocks": 288}),
                   ("576 blocks", {"tn_group_blocks": 576}), ("864 blocks", {"tn_group_blocks": 864})]:
    for k, v in opts.items(): lib.set_option(k, v)
    u1 = timeit(lambda: ops.gemm_tn(a, b, out=out, accumulate=True, colsum=cs))
    u2 = timeit(lambda: ops.conv2_wgrad(dy2, y1, dw, db, accumulate=True))
    print(f"plain  {name:16s} head wgrad {u1:8.1f} us   conv2 wgrad {u2:8.1f} us")
    for k in opts: lib.set_option(k, 0)
