"""config 4 decode alone (bench.decode_rtf_l33: 20 utterances x 5 repeats, forced 36 output steps), device-resident search
against host bookkeeping: python tools/l33_bench.py"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from emoasr_amd.hostenv import respect_cpu_quota
if __import__('os').environ.get('EMOASR_NO_QUOTA') != '1':
    respect_cpu_quota()

dev = torch.device("cuda:0")
for mode in os.environ.get("MODES", "1,0").split(","):
    os.environ["EMOASR_DEVICE_BEAM"] = mode
    with tempfile.TemporaryDirectory() as td:
        r = bench.decode_rtf_l33(dev, torch.bfloat16, td, n_utts=int(os.environ.get("UTTS", 20)), repeats=int(os.environ.get("REPS", 3)))
    print(f"EMOASR_DEVICE_BEAM={mode}: rtf {r['rtf']:.3e}  {r['ms_per_step']:.3f} ms per output step", flush=True)
