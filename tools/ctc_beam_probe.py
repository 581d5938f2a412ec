"""CTC prefix beam search + LM shallow fusion (bench key `ctc_beam`): RTF of the leg alone, LM row cache on / off
(EMOASR_CTC_LM_CACHE), and where the host spends its time."""
import cProfile
import io
import os
import pstats
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace

import torch

import bench
from emoasr_amd.hostenv import respect_cpu_quota
from emoasr_amd.modeling.asr import ASR

respect_cpu_quota()
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ASR(SimpleNamespace(**bench.L2), compute_dtype=torch.bfloat16).to(dev)
with tempfile.TemporaryDirectory() as tmp:
    n = int(os.environ.get("UTTS", 3))
    print(os.environ.get("EMOASR_CTC_LM_CACHE", "1"), bench.ctc_beam_rtf(model, dev, torch.bfloat16, tmp, n_utts=n))
    if os.environ.get("PROFILE"):
        pr = cProfile.Profile()
        pr.enable()
        bench.ctc_beam_rtf(model, dev, torch.bfloat16, tmp, n_utts=1)
        pr.disable()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
        print(s.getvalue())
