import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from types import SimpleNamespace
import bench
from emoasr_amd.modeling.asr import ASR
from emoasr_amd import lib
from emoasr_amd.data import libri_shaped_lengths
dev = torch.device('cuda:0')
torch.manual_seed(2)
model = ASR(SimpleNamespace(**bench.L4), compute_dtype=torch.bfloat16).to(dev).eval()
xlens, _ = libri_shaped_lengths(2000, 0)
rs = np.random.RandomState(4)
pick = rs.choice(len(xlens), 5, replace=False)
utts = [(torch.randn(1, int(xlens[i]), 80).to(dev), [int(xlens[i])]) for i in pick]
for flag in (1, 0):
    lib.set_option("rnnt_greedy_coop", flag)
    model.decode(*utts[0]); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    for x, l in utts:
        h = model.decode(x, l)[0]; n += len(h[0])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("coop", flag, "rtf", el / (sum(l[0] for _, l in utts) * 0.010), "ms/utt", 1e3 * el / 5, "labels", n)
