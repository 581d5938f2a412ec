"""Parameter containers for the encoder blocks.

These classes exist so that `state_dict()` keys, shapes and default initialisation match
the reference checkpoints.  They carry no compute: the HIP engine
(emoasr_amd/engine.py) reads their parameters through the flat arena.  Calling them like
an nn.Module is an error on purpose -- there is no eager/CPU fallback path.

State-dict layout mirrored (reference file:line):
  MultiHeadedAttention / RelMultiHeadedAttention   asr/modeling/transformer.py:48-61, conformer.py:57-66
  PositionwiseFeedForward                          asr/modeling/transformer.py:102-109
  ConvModule                                       asr/modeling/conformer.py:98-119
  ConformerEncoderLayer / TransformerEncoderLayer  conformer.py:146-189, transformer.py:121-142
  Conv2dEncoder                                    asr/modeling/encoders/conv.py:5-19
"""
import torch
import torch.nn as nn


class _Holder(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} is a parameter container; compute runs in emoasr_amd.engine "
                           "(HIP kernels only, no eager fallback)")


class MultiHeadedAttention(_Holder):
    def __init__(self, num_heads, hidden, dropout_rate=0.0, rel=False):
        super().__init__()
        assert hidden % num_heads == 0
        self.h, self.d_k = num_heads, hidden // num_heads
        self.linear_q = nn.Linear(hidden, hidden)
        self.linear_k = nn.Linear(hidden, hidden)
        self.linear_v = nn.Linear(hidden, hidden)
        self.linear_out = nn.Linear(hidden, hidden)
        if rel:
            self.linear_pos = nn.Linear(hidden, hidden, bias=False)
            self.pos_bias_u = nn.Parameter(torch.empty(self.h, self.d_k))
            self.pos_bias_v = nn.Parameter(torch.empty(self.h, self.d_k))
            nn.init.xavier_uniform_(self.pos_bias_u)
            nn.init.xavier_uniform_(self.pos_bias_v)


class PositionwiseFeedForward(_Holder):
    def __init__(self, hidden, inner):
        super().__init__()
        self.w1 = nn.Linear(hidden, inner)
        self.w2 = nn.Linear(inner, hidden)


class ConvModule(_Holder):
    def __init__(self, channels, kernel_size=31):
        super().__init__()
        assert kernel_size % 2 == 1
        self.pointwise_conv1 = nn.Conv1d(channels, 2 * channels, 1)
        self.depthwise_conv = nn.Conv1d(channels, channels, kernel_size, padding=(kernel_size - 1) // 2, groups=channels)
        self.batch_norm = nn.BatchNorm1d(channels)
        self.pointwise_conv2 = nn.Conv1d(channels, channels, 1)


class ConformerEncoderLayer(_Holder):
    def __init__(self, heads, hidden, inner, pos_encode_type="rel"):
        super().__init__()
        self.self_attn = MultiHeadedAttention(heads, hidden, rel=(pos_encode_type == "rel"))
        self.conv = ConvModule(hidden)
        self.feed_forward = PositionwiseFeedForward(hidden, inner)
        self.feed_forward_macaron = PositionwiseFeedForward(hidden, inner)
        self.norm_self_attn = nn.LayerNorm(hidden)
        self.norm_conv = nn.LayerNorm(hidden)
        self.norm_ff = nn.LayerNorm(hidden)
        self.norm_ff_macaron = nn.LayerNorm(hidden)
        self.norm_final = nn.LayerNorm(hidden)


class TransformerEncoderLayer(_Holder):
    def __init__(self, heads, hidden, inner):
        super().__init__()
        self.self_attn = MultiHeadedAttention(heads, hidden)
        self.feed_forward = PositionwiseFeedForward(hidden, inner)
        self.norm1 = nn.LayerNorm(hidden, eps=1e-12)
        self.norm2 = nn.LayerNorm(hidden, eps=1e-12)


class Conv2dEncoder(_Holder):
    def __init__(self, input_dim, output_dim):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(1, output_dim, 3, 2), nn.ReLU(), nn.Conv2d(output_dim, output_dim, 3, 2),
                                  nn.ReLU())
        self.output = nn.Linear(output_dim * (((input_dim - 1) // 2 - 1) // 2), output_dim)


class TransformerDecoderLayer(_Holder):
    """asr/modeling/transformer.py:156-180"""

    def __init__(self, heads, hidden, inner):
        super().__init__()
        self.dec_hidden_size = hidden
        self.self_attn = MultiHeadedAttention(heads, hidden)
        self.src_attn = MultiHeadedAttention(heads, hidden)
        self.feed_forward = PositionwiseFeedForward(hidden, inner)
        self.norm1 = nn.LayerNorm(hidden, eps=1e-12)
        self.norm2 = nn.LayerNorm(hidden, eps=1e-12)
        self.norm3 = nn.LayerNorm(hidden, eps=1e-12)
