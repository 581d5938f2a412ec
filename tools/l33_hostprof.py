"""Host-side profile (cProfile, by internal time) of the config-4 decode leg of bench.py (joint CTC/attention beam 10 + LM)."""
import cProfile, io, os, pstats, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from emoasr_amd.hostenv import respect_cpu_quota
respect_cpu_quota()
dev = torch.device("cuda:0")
with tempfile.TemporaryDirectory() as td:
    bench.decode_rtf_l33(dev, torch.bfloat16, td, n_utts=4, repeats=1)      # warm-up (graphs, lazy set-up)
    pr = cProfile.Profile(); pr.enable()
    r = bench.decode_rtf_l33(dev, torch.bfloat16, td, n_utts=10, repeats=2)
    pr.disable()
print(f"rtf {r['rtf']:.3e}  {r['ms_per_step']:.3f} ms per output step")
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:3600])
