"""CTCForcedAligner -- asr/modeling/decoders/ctc_aligner.py:96-221 on the HIP lattice kernels.

    aligner = CTCForcedAligner(blank_id=0)
    aligns = aligner(log_probs [B,T,V], elens [B], ys [B,L], ylens [B])   -> int64 [B,T] (device)

alpha / beta come from the CTC loss lattice kernel (`emoasr_ctc_forward`), the constrained per-frame arg-max
from `emoasr_ctc_best_path` (csrc/distill.hip).  Unlike the reference, the caller's `log_probs` is not
zero-filled in place beyond elens (ctc_aligner.py:147-150 mutates its argument).
"""
from ..functions import ctc_forced_align_apply


class CTCForcedAligner(object):
    def __init__(self, blank_id=0):
        self.blank_id = blank_id

    def __call__(self, log_probs, elens, ys, ylens):
        return ctc_forced_align_apply(log_probs, elens, ys, ylens, self.blank_id)
