"""ctypes binding of libemoasr_hip.so (the C ABI declared in include/emoasr_hip.h).

The product path has no CPU fallback: if the shared library is missing or an entry
point fails, an exception is raised.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_float, c_int, c_long, c_uint64,
                    c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EMOASR_HIP_LIB") or os.path.join(_HERE, "libemoasr_hip.so")

F32, BF16, F32X3 = 0, 1, 2   # (F32X3: f32 buffers, split-bf16 products -- the mode travels in every call's dtype argument)
ACT_NONE, ACT_RELU, ACT_SWISH, ACT_GELU = 0, 1, 2, 3


class Epilogue(Structure):
    _fields_ = [("bias", c_void_p), ("residual", c_void_p), ("pre_out", c_void_p),
                ("dact_pre", c_void_p), ("alpha", c_float), ("res_scale", c_float),
                ("act", c_int), ("dact", c_int), ("ldr", c_int), ("out_f32", c_int),
                ("drop_p", c_float), ("seed", c_uint64)]


class AttnArgs(Structure):
    _fields_ = [("B", c_int), ("H", c_int), ("DK", c_int), ("Tq", c_int), ("Tk", c_int),
                ("ldq", c_long), ("ldk", c_long), ("ldv", c_long), ("ldo", c_long), ("ldp", c_long),
                ("q", c_void_p), ("k", c_void_p), ("v", c_void_p), ("pos", c_void_p),
                ("bias_u", c_void_p), ("bias_v", c_void_p), ("klens", c_void_p),
                ("causal", c_int), ("scale", c_float), ("drop_p", c_float), ("seed", c_uint64),
                ("out", c_void_p), ("lse", c_void_p),
                ("dout", c_void_p), ("delta", c_void_p),
                ("dq", c_void_p), ("dk", c_void_p), ("dv", c_void_p),
                ("dpos", c_void_p), ("dbias_u", c_void_p), ("dbias_v", c_void_p),
                ("pdT", c_void_p), ("dsT", c_void_p), ("dbd", c_void_p),
                ("ldpd", c_long), ("ldbd", c_long), ("cs", c_void_p),
                ("st", c_void_p), ("ldst", c_long),
                ("qu", c_void_p), ("qv", c_void_p), ("dbias_part", c_void_p),
                ("nseg", c_int), ("seg_b0", c_int * 9), ("seg_T", c_int * 8), ("seg_row", c_long * 9), ("seg_prow", c_long * 9), ("seg_order", c_int * 8),
                ("keep_mask", c_void_p), ("keep_nw", c_int)]


class TnProblem(Structure):
    _fields_ = [("N1", c_int), ("N2", c_int), ("K", c_int), ("A", c_void_p), ("lda", c_long),
                ("B", c_void_p), ("ldb", c_long), ("C", c_void_p), ("ldc", c_long), ("alpha", c_float),
                ("colsum", c_void_p), ("colsum_scale", c_float)]


TN_GROUP_MAX = 16


class LnBwdOpts(Structure):
    _fields_ = [("dy2", c_void_p), ("scale2", c_float), ("drop_p2", c_float), ("seed2", c_uint64),
                ("defer_finalize", c_int)]


class LnFinalizeItem(Structure):
    _fields_ = [("M", c_int), ("N", c_int), ("part", c_void_p), ("dgamma", c_void_p), ("dbeta", c_void_p)]


LN_FINALIZE_MAX = 64


class Lin(Structure):
    _fields_ = [("w", c_void_p), ("b", c_void_p)]


class LnP(Structure):
    _fields_ = [("g", c_void_p), ("b", c_void_p)]


class DecoderLayer(Structure):
    _fields_ = [("ln1", LnP), ("ln2", LnP), ("ln3", LnP), ("qkv", Lin), ("out", Lin), ("q2", Lin), ("out2", Lin),
                ("w1", Lin), ("w2", Lin)]


class DecoderInfer(Structure):
    _fields_ = [("nb", c_int), ("L", c_int), ("T", c_int), ("dd", c_int), ("H", c_int), ("F", c_int), ("V", c_int),
                ("ids", c_void_p), ("embed", c_void_p), ("pe", c_void_p), ("emb_scale", c_float),
                ("kself", c_void_p), ("kmem", c_void_p), ("kv", POINTER(c_void_p)),
                ("ln_out", LnP), ("out", Lin), ("logits_last", c_void_p), ("ws", c_void_p), ("ws_bytes", ctypes.c_size_t)]


class DecoderStep(Structure):
    _fields_ = [("nb", c_int), ("Lmax", c_int), ("T", c_int), ("dd", c_int), ("H", c_int), ("F", c_int), ("V", c_int),
                ("ids", c_void_p), ("pos", c_void_p), ("klens", c_void_p),
                ("embed", c_void_p), ("pe", c_void_p), ("emb_scale", c_float),
                ("kmem", c_void_p), ("kv", POINTER(c_void_p)), ("kcache", c_void_p), ("vcache", c_void_p),
                ("ln_out", LnP), ("out", Lin), ("logits_last", c_void_p), ("ws", c_void_p), ("ws_bytes", ctypes.c_size_t)]


class BertLayer(Structure):
    _fields_ = [("qkv", Lin), ("attn_out", Lin), ("ln_attn", LnP), ("inter", Lin), ("out", Lin), ("ln_out", LnP)]


class BertInfer(Structure):
    _fields_ = [("nb", c_int), ("L", c_int), ("d", c_int), ("H", c_int), ("F", c_int), ("V", c_int),
                ("ids", c_void_p), ("word_emb", c_void_p), ("pe", c_void_p), ("ln_emb", LnP), ("klens", c_void_p),
                ("transform", Lin), ("ln_transform", LnP), ("out_bias", c_void_p), ("logp", c_void_p),
                ("ws", c_void_p), ("ws_bytes", ctypes.c_size_t)]


class BertStep(Structure):
    _fields_ = [("nb", c_int), ("Lmax", c_int), ("d", c_int), ("H", c_int), ("F", c_int), ("V", c_int),
                ("ids", c_void_p), ("pos", c_void_p), ("klens", c_void_p),
                ("word_emb", c_void_p), ("pe", c_void_p), ("ln_emb", LnP), ("kcache", c_void_p), ("vcache", c_void_p),
                ("transform", Lin), ("ln_transform", LnP), ("out_bias", c_void_p), ("logp", c_void_p), ("raw_logits", c_int),
                ("ws", c_void_p), ("ws_bytes", ctypes.c_size_t)]


class BeamUpdate(Structure):
    _fields_ = [("bw", c_int), ("cw", c_int), ("eos", c_int), ("one_minus_lam", c_float), ("lam", c_float), ("mu", c_float),
                ("len_weight", ctypes.c_double), ("vals", c_void_p), ("cands", c_void_p), ("lm_at", c_void_p), ("psi", c_void_p),
                ("score", c_void_p), ("score_ctc", c_void_p),
                ("n_ids", c_void_p), ("n_parent", c_void_p), ("n_pcand", c_void_p), ("n_last", c_void_p),
                ("n_outlen", c_void_p), ("n_klens", c_void_p), ("hist_parent", c_void_p), ("hist_token", c_void_p),
                ("res_score", c_void_p), ("res_step", c_void_p), ("res_parent", c_void_p), ("state", c_void_p),
                ("host_mirror", c_void_p)]


class JointStep(Structure):
    _fields_ = [("dec_nl", c_int), ("dec_layers", POINTER(DecoderLayer)), ("dec", DecoderStep),
                ("lm_nl", c_int), ("lm_layers", POINTER(BertLayer)), ("lm", BertStep),
                ("dec_k_prev", c_void_p), ("dec_v_prev", c_void_p), ("lm_k_prev", c_void_p), ("lm_v_prev", c_void_p),
                ("parent", c_void_p), ("scores_pre", c_void_p), ("ctc_x", c_void_p), ("T", c_int), ("blank", c_int),
                ("states_prev", c_void_p), ("states_cur", c_void_p), ("upd", BeamUpdate)]


class FfnParams(Structure):
    _fields_ = [("ln_g", c_void_p), ("ln_b", c_void_p), ("w1", c_void_p), ("b1", c_void_p),
                ("w2", c_void_p), ("b2", c_void_p)]


class ConformerLayer(Structure):
    _fields_ = [("d", c_int), ("H", c_int), ("F", c_int), ("K", c_int), ("ffm", FfnParams), ("ff", FfnParams),
                ("att_ln_g", c_void_p), ("att_ln_b", c_void_p), ("wqkv", c_void_p), ("bqkv", c_void_p),
                ("wpos", c_void_p), ("bias_u", c_void_p), ("bias_v", c_void_p), ("wout", c_void_p), ("bout", c_void_p),
                ("cv_ln_g", c_void_p), ("cv_ln_b", c_void_p), ("pw1", c_void_p), ("pw1_b", c_void_p),
                ("dw_w", c_void_p), ("dw_b", c_void_p), ("bn_g", c_void_p), ("bn_b", c_void_p),
                ("bn_rm", c_void_p), ("bn_rv", c_void_p), ("bn_nbt", c_void_p), ("pw2", c_void_p), ("pw2_b", c_void_p),
                ("fin_ln_g", c_void_p), ("fin_ln_b", c_void_p),
                ("ffm_w1t", c_void_p), ("ff_w1t", c_void_p), ("wqkv_t", c_void_p), ("pw1_t", c_void_p),
                ("ffm_w2t", c_void_p), ("ff_w2t", c_void_p), ("wout_t", c_void_p), ("pw2_t", c_void_p)]


class TcItem(Structure):
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("rows", c_int), ("cols", c_int), ("ld_dst", c_long)]


TC_MAX = 64


class FfnStash(Structure):
    _fields_ = [("h", c_void_p), ("u", c_void_p), ("a", c_void_p), ("y", c_void_p), ("mean", c_void_p),
                ("rstd", c_void_p)]


MAX_SEGMENTS = 8


class Segments(Structure):
    _fields_ = [("n", c_int), ("B", c_int * MAX_SEGMENTS), ("T", c_int * MAX_SEGMENTS)]


class ConformerFwd(Structure):
    _fields_ = [("B", c_int), ("T", c_int), ("x", c_void_p), ("pos_t", c_void_p), ("klens", c_void_p),
                ("training", c_int), ("p_enc", c_float), ("p_att", c_float), ("seed", c_uint64 * 7),
                ("ffm", FfnStash), ("ff", FfnStash),
                ("at_h", c_void_p), ("qkv", c_void_p), ("pp", c_void_p), ("o", c_void_p), ("at_y", c_void_p),
                ("lse", c_void_p), ("at_mean", c_void_p), ("at_rstd", c_void_p),
                ("cv_h", c_void_p), ("g", c_void_p), ("gl", c_void_p), ("c", c_void_p), ("z", c_void_p),
                ("cv_y", c_void_p), ("bmean", c_void_p), ("bvar", c_void_p), ("bn_part", c_void_p),
                ("cv_mean", c_void_p), ("cv_rstd", c_void_p),
                ("y", c_void_p), ("fin_mean", c_void_p), ("fin_rstd", c_void_p), ("seg", Segments),
                ("att_mask", c_void_p), ("att_mask_nw", c_int), ("att_mask_ready", c_int)]

class ConformerBwd(Structure):
    _fields_ = [("dy", c_void_p), ("dx", c_void_p), ("ws", c_void_p), ("ws_bytes", ctypes.c_size_t),
                ("ln_part", c_void_p), ("ln_part_stride", c_long), ("attn_img", c_void_p), ("attn_img_bytes", ctypes.c_size_t)]


P, I, L, F, U64 = c_void_p, c_int, c_long, c_float, c_uint64

# name -> argtypes (every function returns int status); mirrors include/emoasr_hip.h
SIGNATURES = {
    "emoasr_gemm_nt": [I, I, I, I, P, L, P, L, P, L, POINTER(Epilogue), P],
    "emoasr_gemm_nn": [I, I, I, I, P, L, P, L, P, L, POINTER(Epilogue), P],
    "emoasr_gemm_tn": [I, I, I, I, P, L, P, L, P, L, F, I, P, F, P],
    "emoasr_gemm_nn_batched": [I, I, I, I, P, L, L, L, P, L, L, L, P, L, L, L, I, I, F, I, P],
    "emoasr_gemm_tn_grouped": [I, I, POINTER(TnProblem), P],
    "emoasr_colsum": [I, I, I, P, L, P, F, I, P],
    "emoasr_conv1_fwd": [I, I, I, I, I, P, P, P, P, P],
    "emoasr_conv1_wgrad": [I, I, I, I, I, P, P, P, P, I, P, P],
    "emoasr_conv2_fwd": [I, I, I, I, I, P, P, P, POINTER(Epilogue), P],
    "emoasr_conv2_wgrad": [I, I, I, I, I, P, P, P, P, I, P],
    "emoasr_conv2_col2im": [I, I, I, I, I, P, P, P, P],
    "emoasr_conv2_dgrad": [I, I, I, I, I, P, P, P, P, P],
    "emoasr_conv2_dgrad_kc": [I, I, I, I, I, P, P, P, P, P],
    "emoasr_gemm_nt_big": [I, I, I, I, P, L, P, L, P, L, P, I, P],
    "emoasr_layernorm_fwd": [I, I, I, P, P, P, F, P, P, P, P],
    "emoasr_layernorm_bwd": [I, I, I, P, P, P, P, P, P, P, P, P, P, P],
    "emoasr_layernorm_bwd_ex": [I, I, I, P, P, P, P, P, P, P, P, P, P, POINTER(LnBwdOpts), P],
    "emoasr_layernorm_bwd_finalize": [I, POINTER(LnFinalizeItem), P],
    "emoasr_conformer_layer_fwd": [I, POINTER(ConformerLayer), POINTER(ConformerFwd), P],
    "emoasr_conformer_attn_masks": [I, I, POINTER(Segments), I, I, I, I, P, F, P, P, L, I, P],
    "emoasr_conformer_layer_bwd": [I, POINTER(ConformerLayer), POINTER(ConformerLayer), POINTER(ConformerFwd),
                                   POINTER(ConformerBwd), P],
    "emoasr_transformer_decoder_infer": [I, I, POINTER(DecoderLayer), POINTER(DecoderInfer), P],
    "emoasr_bert_lm_infer": [I, I, POINTER(BertLayer), POINTER(BertInfer), P],
    "emoasr_transformer_decoder_step": [I, I, POINTER(DecoderLayer), POINTER(DecoderStep), P],
    "emoasr_bert_lm_step": [I, I, POINTER(BertLayer), POINTER(BertStep), P],
    "emoasr_beam_cache_gather": [I, I, I, I, I, P, P, P, P, P, P, P],
    "emoasr_beam_update": [POINTER(BeamUpdate), P],
    "emoasr_beam_scores_topk": [I, I, I, I, P, L, P, L, F, P, P, P, P],
    "emoasr_rowlin": [I, I, I, P, L, P, P, F, P, P, I, P, L, P, P, F, P, I, L, P],
    "emoasr_attn_step": [I, I, I, I, P, P, P, P, P, P],
    "emoasr_joint_beam_step": [I, POINTER(JointStep), P, P],
    "emoasr_joint_beam_step_parts": [I, POINTER(JointStep), I, P],
    "emoasr_joint_beam_graph_build": [I, POINTER(JointStep), I, I, P],
    "emoasr_joint_beam_graph_launch": [I, I, P],
    "emoasr_attn_dropmask": [I, POINTER(AttnArgs), P, I, P],
    "emoasr_attn_fwd": [I, POINTER(AttnArgs), P],
    "emoasr_attn_bwd": [I, POINTER(AttnArgs), P],
    "emoasr_attn_bwd_fused": [I, POINTER(AttnArgs), P, ctypes.c_size_t, P],
    "emoasr_glu_fwd": [I, I, I, P, P, P],
    "emoasr_glu_bwd": [I, I, I, P, P, P, P],
    "emoasr_dwconv_fwd": [I, I, I, I, I, P, P, P, P, P],
    "emoasr_dwconv_bwd_x": [I, I, I, I, I, P, P, P, P],
    "emoasr_dwconv_fwd_stats": [I, I, I, I, I, P, P, P, P, P, P],
    "emoasr_bn_stats_finalize": [I, I, I, P, P, P, P, P, F, P, P],
    "emoasr_dwconv_bwd_w": [I, I, I, I, I, P, P, P, P, I, P, P],
    "emoasr_bn_stats": [I, I, I, P, P, P, P, P, F, P],
    "emoasr_bn_swish_fwd": [I, I, I, P, P, P, P, P, F, P, P],
    "emoasr_bn_swish_bwd": [I, I, I, P, P, P, P, P, P, F, P, P, P, P, P],
    "emoasr_glu_dwconv_fwd": [I, I, I, I, I, P, P, P, P, P, P],
    "emoasr_bn_swish_bwd_sums": [I, I, I, P, P, P, P, P, P, F, P, P, P, P, P],
    "emoasr_conv_bwd_fused": [I, I, I, I, I, P, P, P, P, P, P, F, P, P, P, P, P, P, P, P],
    "emoasr_strided_copy": [I, I, P, P, I, I, I, I, L, L, L, L, I, P],
    "emoasr_transpose_cast_batched": [I, I, P, P],
    "emoasr_scale_dropout": [I, L, P, P, F, F, U64, P],
    "emoasr_posenc": [I, I, I, I, P, P, F, F, U64, P, P],
    "emoasr_add": [I, L, P, P, P, P],
    "emoasr_row_lse": [I, I, I, P, L, P, P],
    "emoasr_gemm_nt_lse": [I, I, I, I, P, L, P, L, P, L, P, P, P, P],
    "emoasr_ctc_forward": [I, I, I, I, I, P, L, P, P, P, P, I, P, P, P, P, P],
    "emoasr_ctc_grad": [I, I, I, I, I, P, L, P, P, P, P, I, P, P, P, P, F, P, P, L, P],
    "emoasr_rnnt_greedy": [I, I, I, I, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P, L, P, P, P, P],
    "emoasr_wgrad_side_join": [I, P],
    "emoasr_rnnt_beam_lstm": [I, I, I, I, P, L, P, P, P, P, P, P, P, P, P, P, I, P],
    "emoasr_rnnt_beam_joint": [I, I, I, I, I, P, P, P, P, P, P, P, P],
    "emoasr_rnnt_beam_pick": [I, I, I, I, I, P, L, P, L, P],
    "emoasr_ctc_forward_rows": [I, I, I, I, I, P, L, P, P, P, P, I, P, P, P, P, P, P],
    "emoasr_ctc_grad_rows": [I, I, I, I, I, P, L, P, P, P, P, I, P, P, P, P, F, P, P, P, P, P, L, P],
    "emoasr_ctc_greedy": [I, I, I, I, P, L, P, I, P, P, P, P],
    "emoasr_embed_fwd": [I, I, I, I, P, P, P, F, F, U64, P, P],
    "emoasr_embed_bwd": [I, I, I, P, P, F, F, U64, P, P],
    "emoasr_lsm_loss": [I, I, I, P, L, P, P, F, P, F, P, P, L, P],
    "emoasr_soft_ce": [I, I, I, P, L, P, P, L, P, P, P, P, F, P, F, P, P, L, P],
    "emoasr_ctc_best_path": [I, I, I, P, P, P, P, P, P, I, P, P],
    "emoasr_ctc_label_map": [I, I, P, P, I, I, P, P, P],
    "emoasr_rnnt_best_path": [I, I, I, P, P, P, P, P, P],
    "emoasr_log_softmax": [I, I, I, P, L, P, L, F, P, L, P],
    "emoasr_topk": [I, I, I, P, L, P, L, P, P, P, P],
    "emoasr_ctc_prefix_init": [I, I, P, I, P, P],
    "emoasr_ctc_prefix_score": [I, I, I, I, P, P, I, P, P, P, P, P, P, I, I, P, P, P],
    "emoasr_lstm_cell_fwd": [I, I, I, P, P, P, L, P, P, P],
    "emoasr_lstm_cell_bwd": [I, I, I, P, L, P, P, P, P, P, P, P],
    "emoasr_lstm_seq_fwd": [I, I, I, I, P, P, P, P, P, P, P, P],
    "emoasr_lstm_seq_bwd": [I, I, I, I, P, P, P, P, P, P, P, L, P],
    "emoasr_joint_tanh": [I, I, I, I, I, P, P, P, P],
    "emoasr_joint_reduce": [I, I, I, I, I, P, P, P, P],
    "emoasr_rnnt_forward": [I, I, I, I, I, I, P, P, P, P, I, P, P, P, P, P, P, P],
    "emoasr_rnnt_grad": [I, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, I, F, P, P, P],
    "emoasr_rnnt_head_fwd": [I, L, I, I, I, I, I, I, P, P, P, P, P, I, P, L, P, P, P, P],
    "emoasr_rnnt_ycol": [I, I, I, I, P, P, P, P],
    "emoasr_rnnt_forward_parts": [I, I, I, I, P, P, P, P, P, P, P, P, P, P],
    "emoasr_rnnt_coef": [I, I, I, I, P, P, P, P, P, P, P, P, P, F, P, P, P, P],
    "emoasr_rnnt_head_grad": [I, I, I, I, P, P, P, P, P, I, P, L, P],
    "emoasr_argmax_rows": [I, I, I, P, L, P, P],
    "emoasr_first_not_equal": [I, P, I, P, P],
    "emoasr_sqnorm": [L, P, P, P],
    "emoasr_adam_step": [L, P, P, P, P, F, F, F, F, F, I, P, F, F, P],
    "emoasr_adam_step_ex": [L, P, P, P, P, F, F, F, F, F, I, P, F, F, P, P],
    "emoasr_specaug_apply": [I, I, I, P, P, I, I, P, P, P],
    "emoasr_fbank": [P, L, I, I, I, I, F, P, P, P, I, P],
    "emoasr_cmvn": [I, I, P, P, P, P],
}


class EmoasrHipError(RuntimeError):
    pass


_lib = None


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EmoasrHipError(
            f"{LIB_PATH} not found: build it with `python -m emoasr_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    # torch must load its HIP runtime first: the kernels are enqueued on torch's streams, so this
    # library has to bind to the same libamdhip64 instance (loading ours first gives a second,
    # device-less runtime: "no ROCm-capable device is detected").
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    lib.emoasr_last_error.restype = c_char_p
    lib.emoasr_last_error.argtypes = []
    lib.emoasr_version.restype = c_int
    lib.emoasr_set_option.argtypes = [c_char_p, c_int]
    lib.emoasr_set_option.restype = c_int
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = args
        fn.restype = c_int
    _lib = lib
    # tuning overrides for measurements: EMOASR_OPTIONS="name=value,name=value" (the names of emoasr_set_option)
    for item in filter(None, os.environ.get("EMOASR_OPTIONS", "").split(",")):
        name, _, value = item.partition("=")
        if lib.emoasr_set_option(name.strip().encode(), int(value)) != 0:
            raise EmoasrHipError(f"EMOASR_OPTIONS: {lib.emoasr_last_error().decode()}")
    return lib


_FN = {}

def call(name, *args):
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(load(), name)
    rc = fn(*args)
    if rc != 0:
        raise EmoasrHipError(f"{name} failed ({rc}): {load().emoasr_last_error().decode()}")


_SIZE_FN = {}


def size_query(name, *ints):
    """scratch-size entry points (`long emoasr_*_floats(int, ...)`): the kernels' tiling constants live in
    the library only, callers never restate them"""
    fn = _SIZE_FN.get(name)
    if fn is None:
        fn = getattr(load(), name)
        fn.restype = ctypes.c_long
        fn.argtypes = [c_int] * len(ints)
        _SIZE_FN[name] = fn
    return int(fn(*[int(v) for v in ints]))


def ws_bytes_seg(dtype_code, seg, d, H, F, K):
    """emoasr_conformer_layer_bwd_ws_bytes_seg: backward workspace of a layer pass over stacked micro-batches"""
    fn = load().emoasr_conformer_layer_bwd_ws_bytes_seg
    fn.restype = ctypes.c_size_t
    fn.argtypes = [c_int, POINTER(Segments), c_int, c_int, c_int, c_int]
    n = int(fn(dtype_code, ctypes.byref(seg), d, H, F, K))
    if n == 0:
        raise EmoasrHipError("emoasr_conformer_layer_bwd_ws_bytes_seg: bad segment description")
    return n


def img_bytes_seg(dtype_code, seg, d, H):
    """emoasr_conformer_layer_bwd_img_bytes_seg: zero-filled attention image area of an f32 layer backward (0 for bf16)"""
    fn = load().emoasr_conformer_layer_bwd_img_bytes_seg
    fn.restype = ctypes.c_size_t
    fn.argtypes = [c_int, POINTER(Segments), c_int, c_int]
    return int(fn(dtype_code, ctypes.byref(seg), d, H))


def timer_read(name, reset=True):
    """(launches, summed ms) of a kernel timed inside the library (emoasr_timer_read; option "timers" enables recording)"""
    lib = load()
    calls, ms = c_int(0), ctypes.c_double(0.0)
    lib.emoasr_timer_read.argtypes = [c_char_p, POINTER(c_int), POINTER(ctypes.c_double), c_int]
    lib.emoasr_timer_read.restype = c_int
    if lib.emoasr_timer_read(name.encode(), ctypes.byref(calls), ctypes.byref(ms), int(reset)) != 0:
        raise EmoasrHipError(lib.emoasr_last_error().decode())
    return calls.value, ms.value


TIMER_FAMILIES = ("attn_bwd_fused_kernel", "attn_bwd_dpos_kernel", "attn_fwd_kernel", "gemm_tn_grouped_kernel", "gemm_nt_nn",
                  "gemm_tn", "layernorm", "conv_module")


def timer_mask(*names):
    """value of option "timers" that records only the named families"""
    m = 0
    for n in names:
        m |= 1 << (TIMER_FAMILIES.index(n) + 1)
    return m


def timer_read_ex(name, reset=True):
    """(launches, summed ms, algorithmic flops, algorithmic bytes) of a kernel family timed inside the library"""
    lib = load()
    calls, ms, fl, by = c_int(0), ctypes.c_double(0.0), ctypes.c_double(0.0), ctypes.c_double(0.0)
    fn = lib.emoasr_timer_read_ex
    fn.argtypes = [c_char_p, POINTER(c_int)] + [POINTER(ctypes.c_double)] * 3 + [c_int]
    fn.restype = c_int
    if fn(name.encode(), ctypes.byref(calls), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by), int(reset)) != 0:
        raise EmoasrHipError(lib.emoasr_last_error().decode())
    return calls.value, ms.value, fl.value, by.value


def set_option(name, value):
    lib = load()
    if lib.emoasr_set_option(name.encode(), int(value)) != 0:
        raise EmoasrHipError(lib.emoasr_last_error().decode())
