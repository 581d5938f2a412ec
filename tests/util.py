"""Shared test helpers: golden fixture loading, config objects."""
import os
from types import SimpleNamespace

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

COMMON = dict(input_layer="conv2d", feat_dim=40, num_framestacks=1, enc_hidden_size=128,
              enc_num_attention_heads=2, enc_num_layers=2, enc_intermediate_size=256,
              dropout_enc_rate=0.0, dropout_attn_rate=0.0, dropout_dec_rate=0.0, vocab_size=40,
              blank_id=0, eos_id=2, kd_weight=0, lsm_prob=0.1)
CONFIGS = {
    "l2_tiny": dict(COMMON, encoder_type="conformer", decoder_type="ctc", pos_encode_type="rel"),
    "l1_tiny": dict(COMMON, encoder_type="transformer", decoder_type="ctc"),
}


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    g = {k: torch.from_numpy(z[k]) for k in z.files}
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd/")}
    return SimpleNamespace(**CONFIGS[name]), sd, g


def split_ragged(flat, lens):
    out, o = [], 0
    for n in lens.tolist():
        out.append(flat[o:o + n].tolist())
        o += n
    return out
