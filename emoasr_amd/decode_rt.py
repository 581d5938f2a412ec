"""Python side of the beam-search step runtime (csrc/decode_rt.hip): cached parameter structs for the
Transformer decoder and the Transformer LM, the per-utterance cache of the decoder's cross-attention
keys / values, and one C-ABI call per output step for each network (instead of ~90 and ~100)."""
import ctypes
import math

import torch

from . import lib, ops
from .engine import h2d_i32


def _lin(dst, w, b):
    dst.w, dst.b = w.data_ptr(), (None if b is None else b.data_ptr())


def _ln(dst, g, b):
    dst.g, dst.b = g.data_ptr(), b.data_ptr()


def _ws_bytes(esz, nb, L, d, H, F, V):  # emoasr_decode_ws_bytes
    M = nb * L
    return M * esz * (5 * d + 3 * d + F) + nb * H * L * 4 + nb * V * 4 + nb * d * esz * 2 + 16 * 256


class DecoderStepRuntime:
    """TransformerDecoder.forward_one_step for all live hypotheses of one utterance"""

    def __init__(self, eng):
        self.eng = eng
        self._layers = None
        self._guard = None
        self._ws = None

    def _params(self):
        eng, A = self.eng, self.eng.arena
        A.refresh_shadow()
        guard = (A.flat.data_ptr(), A.shadow.data_ptr())
        if self._layers is not None and self._guard == guard:
            return self._layers
        dd, d = eng.dd, eng.d
        arr = (lib.DecoderLayer * eng.dnl)()
        for li, Ly in enumerate(arr):
            n = f"decoder.transformers.{li}"
            sa, ca, ff = n + ".self_attn", n + ".src_attn", n + ".feed_forward"
            _ln(Ly.ln1, A.p(n + ".norm1.weight"), A.p(n + ".norm1.bias"))
            _ln(Ly.ln2, A.p(n + ".norm2.weight"), A.p(n + ".norm2.bias"))
            _ln(Ly.ln3, A.p(n + ".norm3.weight"), A.p(n + ".norm3.bias"))
            _lin(Ly.qkv, A.w_span(sa + ".linear_q.weight", sa + ".linear_v.weight", (3 * dd, dd)),
                 A.p_span(sa + ".linear_q.bias", sa + ".linear_v.bias", (3 * dd,)))
            _lin(Ly.out, A.w(sa + ".linear_out.weight"), A.p(sa + ".linear_out.bias"))
            _lin(Ly.q2, A.w(ca + ".linear_q.weight"), A.p(ca + ".linear_q.bias"))
            _lin(Ly.out2, A.w(ca + ".linear_out.weight"), A.p(ca + ".linear_out.bias"))
            _lin(Ly.w1, A.w(ff + ".w1.weight"), A.p(ff + ".w1.bias"))
            _lin(Ly.w2, A.w(ff + ".w2.weight"), A.p(ff + ".w2.bias"))
        self._layers, self._guard = arr, guard
        self._F = A.w("decoder.transformers.0.feed_forward.w1.weight").shape[0]
        return arr

    def begin(self, eouts, beam_width):
        """project the utterance's encoder memory to every layer's cross-attention K / V once and replicate it
        for `beam_width` hypotheses: kv[l] is [beam_width, T, 2*dd]"""
        eng, A = self.eng, self.eng.arena
        self._params()
        T, d, dd = eouts.shape[1], eouts.shape[2], eng.dd
        mem = eouts.reshape(T, d)
        self.kv = []
        for li in range(eng.dnl):
            ca = f"decoder.transformers.{li}.src_attn"
            wkv = A.w_span(ca + ".linear_k.weight", ca + ".linear_v.weight", (2 * dd, d))
            bkv = A.p_span(ca + ".linear_k.bias", ca + ".linear_v.bias", (2 * dd,))
            kv1 = ops.gemm_nt(mem, wkv, bias=bkv)  # [T, 2dd]
            self.kv.append(kv1.unsqueeze(0).expand(beam_width, T, 2 * dd).contiguous())
        self.kv_ptrs = (ctypes.c_void_p * eng.dnl)(*[k.data_ptr() for k in self.kv])
        self.T, self.beam = T, beam_width
        self.kmem = h2d_i32([T] * beam_width, eouts.device)

    def step(self, ys_in):
        """ys_in: CPU int64 [nb, L] -> logits of the last position [nb, V] (compute dtype)"""
        eng, A = self.eng, self.eng.arena
        layers = self._params()
        nb, L = ys_in.shape
        assert nb <= self.beam
        dev = A.flat.device
        dd, H, V = eng.dd, eng.dh, A.w("decoder.output.weight").shape[0]
        esz = A.shadow.element_size()
        need = _ws_bytes(esz, nb, L, dd, H, self._F, V)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(max(need, 1 << 22), device=dev, dtype=torch.uint8)
        ids = h2d_i32(ys_in.contiguous(), dev)
        kself = h2d_i32([L] * nb, dev)
        out = torch.empty(nb, V, device=dev, dtype=A.shadow.dtype)
        io = lib.DecoderInfer()
        io.nb, io.L, io.T, io.dd, io.H, io.F, io.V = nb, L, self.T, dd, H, self._F, V
        io.ids, io.embed = ids.data_ptr(), A.w("decoder.embed.weight").data_ptr()
        io.pe, io.emb_scale = eng._abs_table(L, dev, dd).data_ptr(), math.sqrt(dd)
        io.kself, io.kmem = kself.data_ptr(), self.kmem.data_ptr()
        io.kv = ctypes.cast(self.kv_ptrs, ctypes.POINTER(ctypes.c_void_p))
        _ln(io.ln_out, A.p("decoder.norm.weight"), A.p("decoder.norm.bias"))
        _lin(io.out, A.w("decoder.output.weight"), A.p("decoder.output.bias"))
        io.logits_last, io.ws, io.ws_bytes = out.data_ptr(), self._ws.data_ptr(), self._ws.numel()
        lib.call("emoasr_transformer_decoder_infer", ops.dt(out), eng.dnl, layers, ctypes.byref(io), ops._stream())
        self._keep = (ids, kself)  # alive until the next step's kernels are queued behind this one
        return out


class LMStepRuntime:
    """TransformerLM.predict at the last position of equally long prefixes"""

    def __init__(self, lm):
        self.lm = lm
        self._layers = None
        self._guard = None
        self._ws = None

    def _params(self):
        lm = self.lm
        A = lm._bind()
        A.refresh_shadow()
        guard = (A.flat.data_ptr(), A.shadow.data_ptr())
        if self._layers is not None and self._guard == guard:
            return A, self._layers
        P = lm.params
        d = P.hidden_size
        arr = (lib.BertLayer * P.num_layers)()
        pre = "lm.transformer.bert."
        for i, Ly in enumerate(arr):
            lay = f"{pre}encoder.layer.{i}."
            _lin(Ly.qkv, A.w_span(lay + "attention.self.query.weight", lay + "attention.self.value.weight", (3 * d, d)),
                 A.p_span(lay + "attention.self.query.bias", lay + "attention.self.value.bias", (3 * d,)))
            _lin(Ly.attn_out, A.w(lay + "attention.output.dense.weight"), A.p(lay + "attention.output.dense.bias"))
            _ln(Ly.ln_attn, A.p(lay + "attention.output.LayerNorm.weight"), A.p(lay + "attention.output.LayerNorm.bias"))
            _lin(Ly.inter, A.w(lay + "intermediate.dense.weight"), A.p(lay + "intermediate.dense.bias"))
            _lin(Ly.out, A.w(lay + "output.dense.weight"), A.p(lay + "output.dense.bias"))
            _ln(Ly.ln_out, A.p(lay + "output.LayerNorm.weight"), A.p(lay + "output.LayerNorm.bias"))
        self._layers, self._guard = arr, guard
        return A, arr

    def step(self, ys_in):
        """ys_in: CPU int64 [nb, L] (every row complete) -> f32 log-probabilities of the next token [nb, V]"""
        lm = self.lm
        A, layers = self._params()
        P = lm.params
        nb, L = ys_in.shape
        d, H, F, V = P.hidden_size, P.num_attention_heads, P.intermediate_size, P.vocab_size
        dev = A.flat.device
        esz = A.shadow.element_size()
        need = _ws_bytes(esz, nb, L, d, H, F, V)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(max(need, 1 << 22), device=dev, dtype=torch.uint8)
        ids = h2d_i32(ys_in.contiguous(), dev)
        klens = h2d_i32([L] * nb, dev)
        out = torch.empty(nb, V, device=dev, dtype=torch.float32)
        pre, cp = "lm.transformer.bert.", "lm.transformer.cls.predictions."
        io = lib.BertInfer()
        io.nb, io.L, io.d, io.H, io.F, io.V = nb, L, d, H, F, V
        io.ids, io.word_emb, io.pe = ids.data_ptr(), A.w(pre + "embeddings.word_embeddings.weight").data_ptr(), lm._pe.data_ptr()
        _ln(io.ln_emb, A.p(pre + "embeddings.LayerNorm.weight"), A.p(pre + "embeddings.LayerNorm.bias"))
        io.klens = klens.data_ptr()
        _lin(io.transform, A.w(cp + "transform.dense.weight"), A.p(cp + "transform.dense.bias"))
        _ln(io.ln_transform, A.p(cp + "transform.LayerNorm.weight"), A.p(cp + "transform.LayerNorm.bias"))
        io.out_bias, io.logp = A.p(cp + "bias").data_ptr(), out.data_ptr()
        io.ws, io.ws_bytes = self._ws.data_ptr(), self._ws.numel()
        lib.call("emoasr_bert_lm_infer", ops.dt(A.shadow), P.num_layers, layers, ctypes.byref(io), ops._stream())
        self._keep = (ids, klens)
        return out
