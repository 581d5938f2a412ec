"""TransformerDecoder -- module API of asr/modeling/decoders/transformer.py:24-294 on the HIP engine.

    decoder(eouts, elens, eouts_inter, ys, ylens, ys_in, ys_out) -> (loss, loss_dict, logits)
    decoder(eouts, elens, ..., ys_in=ys_in, ys_out=None)         -> logits
    decoder.decode(eouts, elens, eouts_inter, beam_width, len_weight, lm, lm_weight, decode_ctc_weight)
"""
import torch.nn as nn

from ..blocks import TransformerDecoderLayer
from ..functions import attn_decoder_apply, attn_decoder_logits
from .ctc import CTCDecoder


class TransformerDecoder(nn.Module):
    def __init__(self, params, cmlm=False):
        super().__init__()
        if cmlm:
            raise NotImplementedError("emoasr_amd: conditional masked LM decoding is outside the HIP hot path")
        self.vocab_size = params.vocab_size
        self.embed = nn.Embedding(self.vocab_size, params.dec_hidden_size)
        self.dec_num_layers = params.dec_num_layers
        self.transformers = nn.ModuleList(
            TransformerDecoderLayer(params.dec_num_attention_heads, params.dec_hidden_size, params.dec_intermediate_size)
            for _ in range(self.dec_num_layers))
        self.mtl_ctc_weight = params.mtl_ctc_weight
        if self.mtl_ctc_weight > 0:
            self.ctc = CTCDecoder(params, prefix="decoder.ctc")
        self.norm = nn.LayerNorm(params.dec_hidden_size, eps=1e-12)
        self.output = nn.Linear(params.dec_hidden_size, self.vocab_size)
        self.kd_weight = params.kd_weight
        self.blank_id = params.blank_id
        self.eos_id = params.eos_id
        self.max_decode_ylen = params.max_decode_ylen
        self._owner = None

    def forward(self, eouts, elens, eouts_inter=None, ys=None, ylens=None, ys_in=None, ys_out=None,
                soft_labels=None, ps=None, plens=None):
        if ys_out is None:
            return attn_decoder_logits(self, eouts, elens, ys_in, ylens)
        kd = self.kd_weight > 0 and soft_labels is not None  # DistillLoss replaces the label-smoothing loss (:71-79)
        loss, loss_att, loss_ctc, logits, loss_kd = attn_decoder_apply(self, eouts, elens, ys, ylens, ys_in, ys_out,
                                                                       soft_labels if kd else None, self.kd_weight)
        loss_dict = {"loss_kd": loss_kd, "loss_att": loss_att} if kd else {"loss_att": loss_att}
        if self.mtl_ctc_weight > 0:
            loss_dict["loss_ctc"] = loss_ctc
        loss_dict["loss_total"] = loss
        return loss, loss_dict, logits

    def decode(self, eouts, elens, eouts_inter=None, beam_width=1, len_weight=0, lm=None, lm_weight=0,
               decode_ctc_weight=0, decode_phone=False):
        if decode_ctc_weight == 1:
            self.ctc._owner = self._owner
            return self.ctc.decode(eouts, elens, beam_width=1)
        from ..beam_search import joint_beam_search
        return joint_beam_search(self, eouts, elens, beam_width, len_weight, lm, lm_weight, decode_ctc_weight)
