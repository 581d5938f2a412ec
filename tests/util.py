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
    "l2abs_tiny": dict(COMMON, encoder_type="conformer", decoder_type="ctc", pos_encode_type="abs"),
}


L3 = dict(COMMON, encoder_type="conformer", decoder_type="transformer", pos_encode_type="rel",
          dec_hidden_size=128, dec_num_attention_heads=2, dec_num_layers=2, dec_intermediate_size=256,
          mtl_ctc_weight=0.3, loss_normalize_length=False, loss_normalize_batch=True, max_decode_ylen=20)
CONFIGS["l3_tiny"] = L3
L4 = dict(COMMON, encoder_type="conformer", decoder_type="rnn_transducer", pos_encode_type="rel",
          embedding_size=64, dec_hidden_size=128, dec_num_layers=2, joint_hidden_size=128, dropout_emb_rate=0.0,
          mtl_ctc_weight=0.3)
CONFIGS["l4_tiny"] = L4
LM_CFG = dict(lm_type="transformer", vocab_size=40, hidden_size=128, num_layers=2, num_attention_heads=2,
              intermediate_size=256, max_seq_len=64)
DECODE_SETTINGS = [dict(beam_width=4, len_weight=0.0, lm_weight=0.0, decode_ctc_weight=0.0),
                   dict(beam_width=4, len_weight=0.0, lm_weight=0.0, decode_ctc_weight=0.3),
                   dict(beam_width=4, len_weight=0.1, lm_weight=0.3, decode_ctc_weight=0.3),
                   dict(beam_width=3, len_weight=0.2, lm_weight=0.5, decode_ctc_weight=0.0)]


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    g = {k: torch.from_numpy(z[k]) for k in z.files}
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd/")}
    return SimpleNamespace(**CONFIGS[name]), sd, g


def lm_state(g):
    return {k[3:]: v for k, v in g.items() if k.startswith("lm/")}


def split_ragged(flat, lens):
    out, o = [], 0
    for n in lens.tolist():
        out.append(flat[o:o + n].tolist())
        o += n
    return out


CTC_BEAM_SETTINGS = [dict(beam_width=4, len_weight=0.0, lm_weight=0.0),
                     dict(beam_width=4, len_weight=0.1, lm_weight=0.3),
                     dict(beam_width=3, len_weight=0.2, lm_weight=0.5)]


def load_ctc_beam_golden():
    """ctcbeam_tiny: the l2_tiny model with a sharpened output layer + the l3_tiny LM (weights are read
    from those fixtures) and the reference's CTC prefix beam search outputs."""
    cfg, sd, g2 = load_golden("l2_tiny")
    _, _, g3 = load_golden("l3_tiny")
    z = np.load(os.path.join(GOLDEN, "ctcbeam_tiny.npz"))
    gb = {k: torch.from_numpy(z[k]) for k in z.files}
    sd = dict(sd)
    for k, v in gb.items():
        if k.startswith("sd_override/"):
            sd[k[len("sd_override/"):]] = v
    return cfg, sd, lm_state(g3), g2, gb


RNNT_BEAM_WIDTHS = [2, 4]


def load_rnnt_beam_golden():
    """-> {beam_width: [per utterance: list of hyps (each incl. the leading <sos>)]}"""
    g = np.load(os.path.join(GOLDEN, "rnntbeam_tiny.npz"))
    out = {}
    for bw in RNNT_BEAM_WIDTHS:
        flat = split_ragged(g[f"bw{bw}/hyps"], g[f"bw{bw}/hyp_lens"])
        per, k = [], 0
        for n in g[f"bw{bw}/n_hyps"].tolist():
            per.append(flat[k:k + n])
            k += n
        out[bw] = per
    return out


KD_CTC_CASES = {
    "ctc_all": dict(kd_weight=0.5, reduce_main_loss_kd=False),
    "ctc_mid": dict(kd_weight=0.3, reduce_main_loss_kd=True, kd_ctc_soft_label_weight=0.6, kd_ctc_position="mid"),
}
KD_INTER_CASES = {
    "inter": dict(mtl_inter_ctc_weight=0.3, inter_ctc_layer_id=1),
    "inter_kd": dict(mtl_inter_ctc_weight=0.3, inter_ctc_layer_id=1, inter_kd_weight=0.5, kd_weight=0.5,
                     reduce_main_loss_kd=True),
    "inter_kd_noreduce": dict(mtl_inter_ctc_weight=0.3, inter_ctc_layer_id=2, inter_kd_weight=0.5,
                              reduce_main_loss_kd=False),
    "phone": dict(mtl_phone_ctc_weight=0.3, hie_mtl_phone=True, phone_vocab_size=12, inter_ctc_layer_id=1),
    "phone_top": dict(mtl_phone_ctc_weight=0.2, hie_mtl_phone=False, phone_vocab_size=12, inter_ctc_layer_id=1),
}
KD_RNNT_CASES = {"rnnt_word": dict(kd_weight=0.3, kd_type="word", reduce_main_loss_kd=False),
                 "rnnt_word_reduce": dict(kd_weight=0.5, kd_type="word", reduce_main_loss_kd=True)}
KD_ATT = dict(kd_weight=0.4, reduce_main_loss_kd=False)
CAD_CASES = {"cad_all": dict(soft_label_weight=1.0, position="all", lsm_prob=0.1),
             "cad_right": dict(soft_label_weight=0.4, position="right", lsm_prob=0.1),
             "cad_left_nonorm": dict(soft_label_weight=0.0, position="left", lsm_prob=0.2, normalize_length=False,
                                     normalize_batch=False)}
DISTILL_CASES = {"distill": dict(soft_label_weight=0.3, lsm_prob=0.1),
                 "distill_len": dict(soft_label_weight=0.7, lsm_prob=0.0, normalize_length=True, normalize_batch=False)}


def load_kd_golden():
    z = np.load(os.path.join(GOLDEN, "kd_tiny.npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def synthetic_state(shapes, seed=1234):
    """the deterministic initial state make_golden.py gave the reference model (CPU generator, keys in sorted order): the
    fixture holds no weights, both sides build them with this rule"""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros(shp, dtype=torch.int64)
        elif k.endswith("running_mean"):
            sd[k] = torch.zeros(shp)
        elif k.endswith("running_var"):
            sd[k] = torch.ones(shp)
        elif ("norm" in k and k.endswith("weight")):
            sd[k] = 1.0 + 0.05 * torch.randn(shp, generator=g)
        elif len(shp) >= 2:
            fan_in = int(np.prod(shp[1:]))
            sd[k] = torch.randn(shp, generator=g) / fan_in ** 0.5
        else:
            sd[k] = 0.02 * torch.randn(shp, generator=g)
    return sd
