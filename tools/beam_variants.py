"""Device-resident joint beam search against the host bookkeeping over a grid of beam widths, score weights and length limits
(l3_tiny golden weights): f32 hypotheses must be identical, bf16 best hypotheses equal."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
from tests.util import CONFIGS, LM_CFG, load_golden, lm_state
from emoasr_amd.modeling.asr import ASR
from emoasr_amd.modeling.lm import LM
dev = torch.device("cuda:0")
cfg, sd, g = load_golden("l3_tiny")
for dtype in (torch.float32, torch.bfloat16):
    model = ASR(SimpleNamespace(**CONFIGS["l3_tiny"]), compute_dtype=dtype); model.load_state_dict(sd); model = model.to(dev).eval()
    lm = LM(SimpleNamespace(**LM_CFG), compute_dtype=dtype); lm.load_state_dict(lm_state(g)); lm = lm.to(dev).eval()
    bad = 0
    for bw in (1, 2, 7, 20):
        for ctcw, lmw, lw in ((0.0, 0.0, 0.0), (0.3, 0.0, 0.1), (0.5, 0.4, 0.2), (0.0, 0.6, 0.0)):
            for mx in (6, 40):
                model.decoder.max_decode_ylen = mx
                for b in range(2):
                    n = int(g["xlens"][b]); x, xl = g["xs"][b:b+1, :n].to(dev), g["xlens"][b:b+1]
                    kw = dict(beam_width=bw, len_weight=lw, lm=lm, lm_weight=lmw, decode_ctc_weight=ctcw)
                    os.environ["EMOASR_DEVICE_BEAM"] = "1"; h1, s1, _, _ = model.decode(x, xl, **kw)
                    os.environ["EMOASR_DEVICE_BEAM"] = "0"; h0, s0, _, _ = model.decode(x, xl, **kw)
                    same = (h1 == h0) if dtype == torch.float32 else (h1[:1] == h0[:1])
                    if not same or len(s1) != len(s0):
                        bad += 1
                        print("MISMATCH", dtype, bw, ctcw, lmw, lw, mx, b, h1[:2], h0[:2], s1[:2], s0[:2])
    print(dtype, "mismatches:", bad)
