"""CTCDecoder -- module API of asr/modeling/decoders/ctc.py:22-201,346-370 on the HIP engine.

    decoder(eouts, elens, eouts_inter=None, ys=None, ylens=None, ...) -> logits | (loss, loss_dict, logits)
    decoder.decode(eouts, elens, eouts_inter, beam_width, ...) -> (hyps, scores, logits, aligns)
    (beam_width <= 1: greedy; > 1: CTC prefix beam search with optional LM fusion, ctc.py:203-344)
"""
import logging

import torch.nn as nn

from ..functions import ctc_greedy_apply, ctc_head_apply, ctc_loss_apply


class CTCDecoder(nn.Module):
    def __init__(self, params):
        super().__init__()
        self.blank_id = params.blank_id
        self.eos_id = params.eos_id
        self.vocab_size = params.vocab_size
        self.output = nn.Linear(params.enc_hidden_size, self.vocab_size)
        self.kd_weight = params.kd_weight  # read unconditionally, like ctc.py:50
        for key in ("mtl_phone_ctc_weight", "mtl_inter_ctc_weight"):
            if hasattr(params, key) and getattr(params, key) > 0:
                raise NotImplementedError(f"emoasr_amd: {key} > 0 is outside the HIP hot path")
        if self.kd_weight > 0:
            raise NotImplementedError("emoasr_amd: kd_weight > 0 is outside the HIP hot path")
        self._owner = None

    def forward(self, eouts, elens, eouts_inter=None, ys=None, ylens=None, ys_in=None, ys_out=None,
                soft_labels=None, ps=None, plens=None):
        if ys is None:
            return ctc_head_apply(self, eouts)
        loss, logits = ctc_loss_apply(self, eouts, elens, ys, ylens)
        return loss, {"loss_ctc": loss, "loss_total": loss}, logits

    def decode(self, eouts, elens, eouts_inter=None, beam_width=1, len_weight=0, lm=None, lm_weight=0,
               decode_ctc_weight=0, decode_phone=False):
        if beam_width <= 1:
            if lm_weight > 0:
                logging.warning("greedy decoding: LM is not used")
            return ctc_greedy_apply(self, eouts, elens)
        from ..ctc_beam_search import ctc_prefix_beam_search
        hyps, scores, logits = ctc_prefix_beam_search(self, eouts, elens, beam_width, len_weight, lm, lm_weight)
        return hyps, scores, logits, None
