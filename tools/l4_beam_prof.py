"""Host profile of the transducer's alignment-length synchronous beam search (engine.rnnt_beam_search) on one utterance of the
config-5 model: wall time per expansion and where the host spends it."""
import cProfile
import io
import os
import pstats
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from emoasr_amd.hostenv import respect_cpu_quota
from emoasr_amd.modeling.asr import ASR

respect_cpu_quota()
dev = torch.device("cuda:0")
torch.manual_seed(2)
model = ASR(SimpleNamespace(**bench.L4), compute_dtype=torch.bfloat16).to(dev).eval()
frames = int(os.environ.get("FRAMES", 600))
x, l = torch.randn(1, frames, 80).to(dev), [frames]
model.decode(x, l, beam_width=4)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 3
for _ in range(n):
    model.decode(x, l, beam_width=4)
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / n
Tenc = ((frames - 3) // 2 + 1 - 3) // 2 + 1
print(f"{frames} frames ({Tenc} encoder frames): {el * 1e3:.1f} ms per utterance, RTF {el / (frames * 0.010):.2e}, "
      f"{el * 1e6 / (3 * Tenc):.0f} us per expansion round")
pr = cProfile.Profile()
pr.enable()
model.decode(x, l, beam_width=4)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
print(s.getvalue()[:3800])
