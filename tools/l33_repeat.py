"""run bench.py's L3-3 decode measurement a few times in one process (run-to-run spread on one box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
for i in range(3):
    r = bench.decode_rtf_l33(dev, torch.bfloat16)
    print(i, {k: (round(v, 5) if isinstance(v, float) else v) for k, v in r.items() if k in ("rtf", "ms_per_step", "mean_out_steps")}, flush=True)
