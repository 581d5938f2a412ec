"""Transformer LM for shallow fusion -- module API of lm/modeling/lm.py:22-66 and
lm/modeling/transformer.py:19-77 (a causal BERT, vendored HF v3.0.0 modeling_bert.py:159-554),
inference only, on HIP kernels.

    lm = LM(params).cuda();  lm.load_state_dict(reference_lm_state_dict)
    log_probs, states = lm.predict(ys [B,N] int64, ylens [B], states=None)     # [B, V], None

State-dict keys match the reference (`lm.transformer.bert.*`, `lm.transformer.cls.predictions.*`,
output embedding tied to the word embedding).  Only `lm_type == "transformer"` is on the path.
"""
import math
from types import SimpleNamespace
import os

import torch
import torch.nn as nn

from .. import ops
from ..engine import ParamArena, h2d_i32
from ..ops import ACT_GELU
from .blocks import _Holder


class _Self(_Holder):
    def __init__(self, d):
        super().__init__()
        self.query, self.key, self.value = nn.Linear(d, d), nn.Linear(d, d), nn.Linear(d, d)


class _DenseLN(_Holder):
    def __init__(self, din, dout):
        super().__init__()
        self.dense = nn.Linear(din, dout)
        self.LayerNorm = nn.LayerNorm(dout, eps=1e-12)


class _Dense(_Holder):
    def __init__(self, din, dout):
        super().__init__()
        self.dense = nn.Linear(din, dout)


class _Attention(_Holder):
    def __init__(self, d):
        super().__init__()
        self.self = _Self(d)
        self.output = _DenseLN(d, d)


class _Layer(_Holder):
    def __init__(self, d, inner):
        super().__init__()
        self.attention = _Attention(d)
        self.intermediate = _Dense(d, inner)
        self.output = _DenseLN(inner, d)


class _Embeddings(_Holder):
    def __init__(self, vocab, d, max_len):
        super().__init__()
        self.word_embeddings = nn.Embedding(vocab, d, padding_idx=0)
        self.position_embeddings = nn.Embedding(max_len, d)
        self.token_type_embeddings = nn.Embedding(2, d)
        self.LayerNorm = nn.LayerNorm(d, eps=1e-12)


class _Encoder(_Holder):
    def __init__(self, d, inner, n):
        super().__init__()
        self.layer = nn.ModuleList(_Layer(d, inner) for _ in range(n))


class _Bert(_Holder):
    def __init__(self, vocab, d, inner, n, max_len):
        super().__init__()
        self.embeddings = _Embeddings(vocab, d, max_len)
        self.encoder = _Encoder(d, inner, n)
        self.pooler = _Dense(d, d)  # present in reference checkpoints, unused by predict()


class _Predictions(_Holder):
    def __init__(self, vocab, d):
        super().__init__()
        self.transform = _DenseLN(d, d)
        self.decoder = nn.Linear(d, vocab, bias=False)
        self.bias = nn.Parameter(torch.zeros(vocab))
        self.decoder.bias = self.bias


class _Cls(_Holder):
    def __init__(self, vocab, d):
        super().__init__()
        self.predictions = _Predictions(vocab, d)


class _BertForMaskedLM(_Holder):
    def __init__(self, vocab, d, inner, n, max_len):
        super().__init__()
        self.bert = _Bert(vocab, d, inner, n, max_len)
        self.cls = _Cls(vocab, d)
        self.cls.predictions.decoder.weight = self.bert.embeddings.word_embeddings.weight  # tied


class TransformerLM(nn.Module):
    def __init__(self, params):
        super().__init__()
        self.transformer = _BertForMaskedLM(params.vocab_size, params.hidden_size, params.intermediate_size,
                                            params.num_layers, params.max_seq_len)

    def zero_states(self, bs, device):
        return None  # stateless


class LM(nn.Module):
    def __init__(self, params, phase="test", compute_dtype=torch.bfloat16):
        super().__init__()
        self.lm_type = params.lm_type
        if self.lm_type != "transformer":
            raise NotImplementedError(f"emoasr_amd: lm_type={self.lm_type!r} is outside the HIP hot path")
        self.params = params
        # "f32x3" (f32 storage, split-bf16 products: modeling/asr.py) -> float32 + the library's split switch asserted per call
        self.f32_split = isinstance(compute_dtype, str) and compute_dtype == "f32x3"
        self.compute_dtype = torch.float32 if self.f32_split else compute_dtype
        self.lm = TransformerLM(params)
        self._arena = None
        self._pe = None

    def load_state_dict(self, state_dict, strict=True):
        try:
            return super().load_state_dict(state_dict, strict)
        except RuntimeError:
            return self.lm.load_state_dict(state_dict, strict)  # un-prefixed inner dict (lm.py:62-66)

    def zero_states(self, bs, device):
        return self.lm.zero_states(bs, device)

    # ---------------------------------------------------------------- HIP forward
    def _bind(self):
        if self._arena is None or not self._arena.bound() or self._arena.compute_dtype != self.compute_dtype:
            self._arena = ParamArena(self, self.compute_dtype)
            emb = "lm.transformer.bert.embeddings."
            A = self._arena
            # position + token-type(0) rows folded into one additive table (modeling_bert.py:196-201)
            self._pe = (A.p(emb + "position_embeddings.weight") + A.p(emb + "token_type_embeddings.weight")[0]).contiguous()
        return self._arena

    def _forward_rows(self, ids, klens, idx, B, N):
        """ids int32 [B,N], klens int32 [B], idx [B] (flat position b * N + ylens[b] - 1 of every row's last token), all on the
        device -> f32 log-probs [B, V]"""
        A = self._arena
        P = self.params
        d, H, nl = P.hidden_size, P.num_attention_heads, P.num_layers
        pre = "lm.transformer.bert."
        x = ops.embed_fwd(ids, A.w(pre + "embeddings.word_embeddings.weight"), self._pe, 1.0).view(B * N, d)
        x, _, _ = ops.layernorm_fwd(x, A.p(pre + "embeddings.LayerNorm.weight"), A.p(pre + "embeddings.LayerNorm.bias"),
                                    1e-12, False)
        scale = 1.0 / math.sqrt(d // H)
        for i in range(nl):
            lay = f"{pre}encoder.layer.{i}."
            wqkv = A.w_span(lay + "attention.self.query.weight", lay + "attention.self.value.weight", (3 * d, d))
            bqkv = A.p_span(lay + "attention.self.query.bias", lay + "attention.self.value.bias", (3 * d,))
            qkv = ops.gemm_nt(x, wqkv, bias=bqkv).view(B, N, 3 * d)
            o, _ = ops.attn_fwd(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], H, scale, klens=klens, causal=True)
            y = ops.gemm_nt(o.view(B * N, d), A.w(lay + "attention.output.dense.weight"),
                            bias=A.p(lay + "attention.output.dense.bias"), residual=x, res_scale=1.0)
            x1, _, _ = ops.layernorm_fwd(y, A.p(lay + "attention.output.LayerNorm.weight"),
                                         A.p(lay + "attention.output.LayerNorm.bias"), 1e-12, False)
            u = ops.gemm_nt(x1, A.w(lay + "intermediate.dense.weight"), bias=A.p(lay + "intermediate.dense.bias"),
                            act=ACT_GELU)
            y2 = ops.gemm_nt(u, A.w(lay + "output.dense.weight"), bias=A.p(lay + "output.dense.bias"), residual=x1,
                             res_scale=1.0)
            x, _, _ = ops.layernorm_fwd(y2, A.p(lay + "output.LayerNorm.weight"), A.p(lay + "output.LayerNorm.bias"),
                                        1e-12, False)
        rows = x.index_select(0, idx)
        cp = "lm.transformer.cls.predictions."
        t = ops.gemm_nt(rows, A.w(cp + "transform.dense.weight"), bias=A.p(cp + "transform.dense.bias"), act=ACT_GELU)
        t, _, _ = ops.layernorm_fwd(t, A.p(cp + "transform.LayerNorm.weight"), A.p(cp + "transform.LayerNorm.bias"),
                                    1e-12, False)
        logits = ops.gemm_nt(t, A.w(pre + "embeddings.word_embeddings.weight"), bias=A.p(cp + "bias"), out_f32=True)
        return ops.log_softmax(logits)

    def predict_device(self, ys, ylens):
        """ys CPU int64 [B,N]; ylens list -> f32 log-probs [B, V] on the device (row ylens[b]-1).

        Up to 16 rows (what the beam searches ask for, one call per output step or per frame): the ~90 launches of the forward
        are replayed from a HIP graph over static buffers, one graph per (rows padded to 4 / 16, length padded to a multiple of
        8) -- the call was host-bound (1.5 ms of launch sequencing for ~0.4 ms of kernels).  Padding rows / positions are masked
        by their key lengths and never read back; the real rows' arithmetic is the eager call's.  EMOASR_LM_GRAPH=0: eager."""
        arena_before = self._arena
        A = self._bind()
        if A is not arena_before:
            self._graphs = {}     # (captured launches hold the old arena's addresses)
        A.refresh_shadow()
        ys = torch.as_tensor(ys)
        B, N = ys.shape
        dev = A.flat.device
        yl = [int(v) for v in ylens]
        split = self.f32_split if self.compute_dtype == torch.float32 else None
        if B <= 16 and dev.type == "cuda" and os.environ.get("EMOASR_LM_GRAPH", "1") != "0" and not torch.cuda.is_current_stream_capturing():
            return self._predict_graph(ys, yl, B, N, dev, split)
        with ops.stream_scope(split):
            ids = h2d_i32(ys.contiguous(), dev)
            klens = h2d_i32(yl, dev)
            idx = h2d_i32([b * N + v - 1 for b, v in enumerate(yl)], dev)
            return self._forward_rows(ids, klens, idx, B, N)

    def _predict_graph(self, ys, yl, B, N, dev, split):
        Bp, Np = (4 if B <= 4 else 16), (N + 7) // 8 * 8
        graphs = self.__dict__.setdefault("_graphs", {})
        g = graphs.get((Bp, Np))
        if g is None:
            g = SimpleNamespace()
            n_ids = Bp * Np
            g.host = torch.zeros(n_ids + 2 * Bp, dtype=torch.int32).pin_memory()
            g.dev = torch.zeros(n_ids + 2 * Bp, dtype=torch.int32, device=dev)
            g.ids, g.klens, g.idx = g.dev[:n_ids].view(Bp, Np), g.dev[n_ids:n_ids + Bp], g.dev[n_ids + Bp:]
            g.host[n_ids:n_ids + Bp] = 1     # (a valid state for the warm-up run: every row one token long)
            g.host[n_ids + Bp:] = torch.arange(Bp, dtype=torch.int32) * Np
            g.dev.copy_(g.host)

            def body():
                with ops.stream_scope(split):
                    return self._forward_rows(g.ids, g.klens, g.idx, Bp, Np)

            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                body()   # warm-up outside the capture (allocator, lazy initialisation, per-stream scratch)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g.graph):
                g.out = body()
            graphs[(Bp, Np)] = g
        n_ids = Bp * Np
        h = g.host
        if getattr(g, "ev", None) is not None:
            g.ev.synchronize()   # (the previous call's upload has left the pinned record)
        h[:n_ids + Bp] = 0
        h[:n_ids].view(Bp, Np)[:B, :N] = ys.to(torch.int32)
        h[n_ids:n_ids + Bp] = 1
        h[n_ids:n_ids + B] = torch.tensor(yl, dtype=torch.int32)
        h[n_ids + Bp:] = torch.arange(Bp, dtype=torch.int32) * Np
        h[n_ids + Bp:n_ids + Bp + B] += torch.tensor(yl, dtype=torch.int32) - 1
        g.dev.copy_(h, non_blocking=True)
        g.ev = torch.cuda.Event()
        g.ev.record()
        g.graph.replay()
        return g.out[:B].clone()

    def predict(self, ys, ylens, states=None):
        with torch.no_grad():
            ys_host = ys.cpu() if torch.is_tensor(ys) else torch.as_tensor(ys)
            yl = ylens.tolist() if torch.is_tensor(ylens) else list(ylens)
            return self.predict_device(ys_host, yl), states
