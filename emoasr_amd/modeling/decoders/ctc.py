"""CTCDecoder -- module API of asr/modeling/decoders/ctc.py:22-201,346-370 on the HIP engine.

    decoder(eouts, elens, eouts_inter=None, ys=None, ylens=None, ...) -> logits | (loss, loss_dict, logits)
    decoder.decode(eouts, elens, eouts_inter, beam_width, ...) -> (hyps, scores, logits, aligns)
    (beam_width <= 1: greedy; > 1: CTC prefix beam search with optional LM fusion, ctc.py:203-344)

With every auxiliary weight at 0 (the L-series configs) the head GEMM, CTC lattice and gradient run as one
fused autograd node.  Knowledge distillation (kd_weight, ctc.py:117-127), the phone-level CTC
(mtl_phone_ctc_weight, :129-148) and the intermediate CTC (mtl_inter_ctc_weight [+ inter_kd_weight],
:150-170) compose separate nodes: head -> logits, logits -> CTC loss (+ forced alignment from the same
lattices), logits -> CTCAlignDistillLoss.
"""
import logging

import torch.nn as nn

from ...criteria import CTCAlignDistillLoss
from ..functions import ctc_from_logits_apply, ctc_greedy_apply, ctc_head_apply, ctc_loss_apply, head_apply
from .ctc_aligner import CTCForcedAligner


def _opt(params, key, default=0):
    return getattr(params, key) if hasattr(params, key) else default


class CTCDecoder(nn.Module):
    def __init__(self, params, prefix="decoder"):
        super().__init__()
        self.blank_id = params.blank_id
        self.eos_id = params.eos_id
        self.vocab_size = params.vocab_size
        self.output = nn.Linear(params.enc_hidden_size, self.vocab_size)
        self.mtl_phone_ctc_weight = _opt(params, "mtl_phone_ctc_weight")
        self.mtl_inter_ctc_weight = _opt(params, "mtl_inter_ctc_weight")
        self.kd_weight = params.kd_weight  # read unconditionally, like ctc.py:50
        if self.mtl_phone_ctc_weight > 0:
            self.hie_mtl_phone = params.hie_mtl_phone
            self.phone_output = nn.Linear(params.enc_hidden_size, params.phone_vocab_size)

        def kd_loss():
            return CTCAlignDistillLoss(vocab_size=params.vocab_size, blank_id=params.blank_id, lsm_prob=params.lsm_prob,
                                       soft_label_weight=_opt(params, "kd_ctc_soft_label_weight", 1.0),
                                       position=_opt(params, "kd_ctc_position", "all"))

        if self.kd_weight > 0:
            self.ctc_kd_loss_fn = kd_loss()
            self.reduce_main_loss_kd = params.reduce_main_loss_kd
            self.forced_aligner = CTCForcedAligner(blank_id=self.blank_id)
        if self.mtl_inter_ctc_weight > 0:
            self.inter_kd_weight = _opt(params, "inter_kd_weight")
            if self.inter_kd_weight > 0:
                self.inter_ctc_kd_loss_fn = kd_loss()
                self.reduce_main_loss_kd = params.reduce_main_loss_kd
                self.forced_aligner = CTCForcedAligner(blank_id=self.blank_id)
        self._prefix = prefix  # parameter-name prefix of this module inside the owning ASR model
        self._owner = None

    def forward(self, eouts, elens, eouts_inter=None, ys=None, ylens=None, ys_in=None, ys_out=None,
                soft_labels=None, ps=None, plens=None):
        if ys is None:
            return ctc_head_apply(self, eouts, self._prefix + ".output")
        kd = self.kd_weight > 0 and soft_labels is not None
        if not (kd or self.mtl_phone_ctc_weight > 0 or self.mtl_inter_ctc_weight > 0):
            loss, logits = ctc_loss_apply(self, eouts, elens, ys, ylens, self._prefix + ".output")
            return loss, {"loss_ctc": loss, "loss_total": loss}, logits
        loss_dict = {}
        head = self._prefix + ".output"
        logits = head_apply(self, eouts, head)
        loss_ctc, aligns = ctc_from_logits_apply(self, logits, eouts, elens, ys, ylens, want_aligns=kd)
        loss = loss_ctc
        loss_dict["loss_ctc"] = loss_ctc
        if kd:
            loss_kd = self.ctc_kd_loss_fn(logits, ys, soft_labels, aligns, elens, ylens)
            loss_dict["loss_kd"] = loss_kd
            if self.reduce_main_loss_kd:
                loss = (1 - self.kd_weight) * loss + self.kd_weight * loss_kd
            else:
                loss = loss + self.kd_weight * loss_kd
        if self.mtl_phone_ctc_weight > 0:
            src = eouts_inter if self.hie_mtl_phone else eouts  # hierarchical: intermediate layer (ctc.py:133-137)
            logits_phone = head_apply(self, src, self._prefix + ".phone_output")
            loss_phone, _ = ctc_from_logits_apply(self, logits_phone, eouts, elens, ps, plens)
            loss = loss + self.mtl_phone_ctc_weight * loss_phone
            loss_dict["loss_phone_ctc(inter)" if self.hie_mtl_phone else "loss_phone_ctc"] = loss_phone
        if self.mtl_inter_ctc_weight > 0:
            inter_kd = self.inter_kd_weight > 0  # (the reference does not test soft_labels here, ctc.py:159)
            logits_inter = head_apply(self, eouts_inter, head)
            loss_inter, aligns_inter = ctc_from_logits_apply(self, logits_inter, eouts, elens, ys, ylens,
                                                             want_aligns=inter_kd)
            loss_dict["loss_inter_ctc"] = loss_inter
            if inter_kd:
                loss_inter_kd = self.inter_ctc_kd_loss_fn(logits_inter, ys, soft_labels, aligns_inter, elens, ylens)
                loss_dict["loss_inter_kd"] = loss_inter_kd
                if self.reduce_main_loss_kd:
                    loss = loss + self.mtl_inter_ctc_weight * ((1 - self.inter_kd_weight) * loss_inter
                                                               + self.inter_kd_weight * loss_inter_kd)
                else:
                    loss = loss + self.inter_kd_weight * loss_inter_kd  # (ctc.py:167: the inter CTC loss is dropped)
            else:
                loss = loss + self.mtl_inter_ctc_weight * loss_inter
        loss_dict["loss_total"] = loss
        return loss, loss_dict, logits

    def decode(self, eouts, elens, eouts_inter=None, beam_width=1, len_weight=0, lm=None, lm_weight=0,
               decode_ctc_weight=0, decode_phone=False):
        if decode_phone:
            raise NotImplementedError("emoasr_amd: decode_phone (broken in the reference, test_asr.py:222) is not provided")
        if beam_width <= 1:
            if lm_weight > 0:
                logging.warning("greedy decoding: LM is not used")
            return ctc_greedy_apply(self, eouts, elens)
        from ..ctc_beam_search import ctc_prefix_beam_search
        hyps, scores, logits = ctc_prefix_beam_search(self, eouts, elens, beam_width, len_weight, lm, lm_weight)
        return hyps, scores, logits, None
