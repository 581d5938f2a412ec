"""config 4, the <eos>-biased leg (hypotheses finish, the search stops by itself): per-utterance timing of the search loop with the
host-side split (EMOASR_BEAM_TIMING / EMOASR_BEAM_STEP_TIMING): python tools/l33_eos_probe.py"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from emoasr_amd.hostenv import respect_cpu_quota
respect_cpu_quota()
dev = torch.device("cuda:0")
os.environ["EMOASR_BEAM_TIMING"] = os.environ.get("EMOASR_BEAM_TIMING", "1")
os.environ["EMOASR_BEAM_STEP_TIMING"] = "1"
with tempfile.TemporaryDirectory() as td:
    r = bench.decode_rtf_l33(dev, torch.bfloat16, td, n_utts=int(os.environ.get("UTTS", 10)), repeats=2, eos_biased=True)
print(r)
