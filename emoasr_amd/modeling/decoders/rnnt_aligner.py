"""RNNTForcedAligner -- asr/modeling/decoders/rnnt_aligner.py:158-198 on the HIP transducer lattice kernels.

    aligner = RNNTForcedAligner(blank_id=0)
    aligns = aligner(log_probs [B,T,L+1,V], elens [B], ys [B,L], ylens [B])   -> int32 [B,L] (device)

alpha / beta come from the transducer loss lattice kernel (`emoasr_rnnt_forward`; the reference runs its
own Numba CUDA kernels, :14-152), the greedy alpha+beta walk from `emoasr_rnnt_best_path`.
"""
from ..functions import rnnt_forced_align_apply


class RNNTForcedAligner(object):
    def __init__(self, blank_id=0):
        self.blank_id = blank_id

    def __call__(self, log_probs, elens, ys, ylens):
        return rnnt_forced_align_apply(log_probs, elens, ys, ylens, self.blank_id)
