"""TransformerEncoder (Transformer and Conformer stacks) -- module API of
asr/modeling/encoders/transformer.py:16-113 on the HIP engine.

    encoder(xs, xlens) -> (eouts [B,T',d], elens [B] int64, eouts_inter | None)
"""
import torch
import torch.nn as nn

from ..blocks import Conv2dEncoder, ConformerEncoderLayer, TransformerEncoderLayer
from ..functions import encoder_apply


class TransformerEncoder(nn.Module):
    def __init__(self, params, is_conformer=False):
        super().__init__()
        self.params = params
        self.input_layer = params.input_layer
        if self.input_layer != "conv2d":
            raise NotImplementedError("emoasr_amd: only input_layer='conv2d' is on the HIP path")
        self.enc_num_layers = params.enc_num_layers
        self.pos_encode_type = params.pos_encode_type if hasattr(params, "pos_encode_type") else "abs"
        self.is_conformer = is_conformer
        if self.pos_encode_type == "rel":
            assert is_conformer
        d = params.enc_hidden_size
        self.conv = Conv2dEncoder(params.feat_dim * params.num_framestacks, d)
        self.transformers = nn.ModuleList()
        for _ in range(self.enc_num_layers):
            if is_conformer:
                layer = ConformerEncoderLayer(params.enc_num_attention_heads, d, params.enc_intermediate_size,
                                              self.pos_encode_type)
            else:
                layer = TransformerEncoderLayer(params.enc_num_attention_heads, d, params.enc_intermediate_size)
            self.transformers.append(layer)
        self.norm = nn.LayerNorm(d, eps=1e-12)
        inter = (hasattr(params, "mtl_inter_ctc_weight") and params.mtl_inter_ctc_weight > 0) or \
                (hasattr(params, "mtl_phone_ctc_weight") and params.mtl_phone_ctc_weight > 0)
        self.inter_ctc_layer_id = params.inter_ctc_layer_id if inter else 0
        self._owner = None  # set by ASR so encoder and decoder share one engine / arena

    def forward(self, xs, xlens):
        return encoder_apply(self, xs, xlens)  # (eouts, elens, eouts_inter | None)
