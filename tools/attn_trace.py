import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("B", "20"); os.environ.setdefault("T", "374")
import numpy as np, torch
import runpy
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "attn_bench.py"))
from emoasr_amd import lib
buf = (ctypes.c_ulonglong * (2048 * 16))()
L = lib.load(); L.emoasr_debug_atrace.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.emoasr_debug_atrace(buf, 2048 * 16)
t = np.array(buf[:], dtype=np.float64).reshape(2048, 16)[:, :10]
t = t[(t[:, 0] > 0) & (t[:, 9] > t[:, 0])]
d = np.diff(t, axis=1)
names = ["regs->LDS", "dP mfma", "fetch issue", "S+band+skew", "elementwise+img", "pdT/dsT stores", "dQu mfma", "dBD stores", "dQv"]
print("blocks", len(t), "median cycles per phase of one key tile (wave 0, second tile):")
for n, v, m in zip(names, np.median(d, axis=0), d.mean(axis=0)): print(f"  {n:18s} {v:8.0f}  (mean {m:8.0f})")
print("  total", np.median(t[:, 9] - t[:, 0]))
