import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emoasr_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
x = torch.randn(136800, 2304, device=dev).bfloat16(); y = torch.empty_like(x)
nbytes = x.numel() * 2
t = timeit(lambda: ops.scale_dropout(x, 1.0, 0.0, 0)); print(f"scale_dropout 630MB->630MB: {t:.1f} us  {2 * nbytes / t / 1e6:.2f} TB/s")
t = timeit(lambda: y.copy_(x)); print(f"torch copy_: {t:.1f} us  {2 * nbytes / t / 1e6:.2f} TB/s")
t = timeit(lambda: y.zero_()); print(f"torch zero_ (write only): {t:.1f} us  {nbytes / t / 1e6:.2f} TB/s")
t = timeit(lambda: y.fill_(1.0)); print(f"torch fill_ (write only): {t:.1f} us  {nbytes / t / 1e6:.2f} TB/s")
