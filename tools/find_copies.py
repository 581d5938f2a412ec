"""Which call sites issue torch copy / fill ops in one training step?  (monkeypatched Tensor methods)"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--steps", "2", "--warmup", "2", "--no-cpu-baseline", "--no-decode"]
import torch
from collections import Counter
cnt = Counter()
def wrap(obj, name):
    orig = getattr(obj, name)
    def f(*a, **k):
        for fr in reversed(traceback.extract_stack(limit=8)[:-1]):
            if "emoasr_amd" in fr.filename or fr.filename.endswith("bench.py"):
                cnt[(name, fr.filename.split("/root/repo/")[-1].split("/")[-1], fr.lineno, fr.line)] += 1
                break
        return orig(*a, **k)
    setattr(obj, name, f)
for n in ("to", "copy_", "fill_", "zero_", "clone", "contiguous", "float", "add_", "mul_", "sum", "__getitem__", "__setitem__"):
    wrap(torch.Tensor, n)
for n in ("zeros", "zeros_like", "tensor", "as_tensor", "where", "cat", "stack"):
    wrap(torch, n)
import bench
bench.main()
for (name, f, ln, line), n in cnt.most_common(60):
    print(n, name, f, ln, (line or "")[:90], file=sys.stderr)
