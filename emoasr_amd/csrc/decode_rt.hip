// Beam-search step runtime: the Transformer decoder (whole prefix, cached cross-attention K/V) and the
// BERT-style Transformer LM as ONE C-ABI call each per output step -- the same kernels the Python
// sequencing (engine.py:_dec_forward, modeling/lm.py:predict_device) launches one FFI call at a time,
// minus the work the beam search never uses (logits of earlier positions, K/V of the unchanged memory).
// Reference: decoders/transformer.py:148-159 + transformer.py:156-198; lm/modeling/transformer.py:62-77.
#include <math.h>
#include "common.h"
#include "../../include/emoasr_hip.h"

namespace {

struct Bump {  // carve 256-byte aligned pieces out of the caller's scratch
  char* p; size_t left; bool ok = true;
  void* take(size_t n) {
    n = (n + 255) / 256 * 256;
    if (n > left) { ok = false; return nullptr; }
    void* r = p; p += n; left -= n; return r;
  }
};

emoasr_epilogue_t ep0() { emoasr_epilogue_t e{}; e.alpha = 1.f; e.res_scale = 1.f; return e; }

int linear(int dtype, int M, int N, int K, const void* x, long ldx, const emoasr_lin_t& l, void* y, int act,
           const void* residual, void* stream, int out_f32 = 0) {
  emoasr_epilogue_t e = ep0();
  e.bias = l.b; e.act = act; e.residual = residual; e.ldr = N; e.out_f32 = out_f32;
  return emoasr_gemm_nt(dtype, M, N, K, x, ldx, l.w, K, y, N, &e, stream);
}

int self_attn(int dtype, int nb, int L, int d, int H, const void* qkv, const int* klens, int causal, void* o, float* lse,
              void* stream) {
  const size_t esz = dtype == EMO_BF16 ? 2 : 4;
  emoasr_attn_t a{};
  a.B = nb; a.H = H; a.DK = d / H; a.Tq = L; a.Tk = L;
  a.ldq = a.ldk = a.ldv = 3 * d; a.ldo = d;
  a.q = qkv; a.k = (const char*)qkv + (size_t)d * esz; a.v = (const char*)qkv + (size_t)2 * d * esz;
  a.klens = klens; a.causal = causal; a.scale = 1.f / sqrtf((float)(d / H));
  a.out = o; a.lse = lse;
  return emoasr_attn_fwd(dtype, &a, stream);
}

}  // namespace

extern "C" size_t emoasr_decode_ws_bytes(int dtype, int nb, int L, int d, int H, int F, int V) {
  const size_t esz = dtype == EMO_BF16 ? 2 : 4, M = (size_t)nb * L;
  // x, x', h, o, q (d each), qkv (3d), a (F), lse, logits (V per hypothesis, f32 at most) + alignment slack
  return M * esz * ((size_t)5 * d + 3 * d + F) + (size_t)nb * H * L * 4 + (size_t)nb * V * 4 + (size_t)nb * d * esz * 2 +
         16 * 256;
}

extern "C" int emoasr_transformer_decoder_infer(int dtype, int nl, const emoasr_decoder_layer_t* layers,
                                                const emoasr_decoder_infer_t* io, void* stream) {
  EMO_CHECK(layers && io && io->ws && io->kv && io->logits_last, "decoder_infer: missing arguments");
  const int nb = io->nb, L = io->L, T = io->T, dd = io->dd, H = io->H, F = io->F, M = nb * L;
  EMO_CHECK(nb > 0 && L > 0 && T > 0 && dd % H == 0, "decoder_infer: bad dims nb=%d L=%d T=%d dd=%d H=%d", nb, L, T, dd, H);
  const size_t esz = dtype == EMO_BF16 ? 2 : 4;
  Bump ws{(char*)io->ws, io->ws_bytes};
  void* x = ws.take(M * dd * esz);
  void* x2 = ws.take(M * dd * esz);
  void* h = ws.take(M * dd * esz);
  void* o = ws.take(M * dd * esz);
  void* q = ws.take(M * dd * esz);
  void* qkv = ws.take((size_t)M * 3 * dd * esz);
  void* act = ws.take((size_t)M * F * esz);
  float* lse = (float*)ws.take((size_t)nb * H * L * 4);
  EMO_CHECK(ws.ok, "decoder_infer: scratch too small (%zu bytes given)", io->ws_bytes);
  if (emoasr_embed_fwd(dtype, M, L, dd, io->ids, io->embed, io->pe, io->emb_scale, 0.f, 0, x, stream)) return 1;
  for (int li = 0; li < nl; ++li) {
    const emoasr_decoder_layer_t& Ly = layers[li];
    // masked self-attention over the prefix
    if (emoasr_layernorm_fwd(dtype, M, dd, x, Ly.ln1.g, Ly.ln1.b, 1e-12f, h, nullptr, nullptr, stream)) return 1;
    if (linear(dtype, M, 3 * dd, dd, h, dd, Ly.qkv, qkv, EMOASR_ACT_NONE, nullptr, stream)) return 1;
    if (self_attn(dtype, nb, L, dd, H, qkv, io->kself, 1, o, lse, stream)) return 1;
    if (linear(dtype, M, dd, dd, o, dd, Ly.out, x2, EMOASR_ACT_NONE, x, stream)) return 1;
    // source attention against the cached memory projections
    if (emoasr_layernorm_fwd(dtype, M, dd, x2, Ly.ln2.g, Ly.ln2.b, 1e-12f, h, nullptr, nullptr, stream)) return 1;
    if (linear(dtype, M, dd, dd, h, dd, Ly.q2, q, EMOASR_ACT_NONE, nullptr, stream)) return 1;
    {
      emoasr_attn_t a{};
      a.B = nb; a.H = H; a.DK = dd / H; a.Tq = L; a.Tk = T;
      a.ldq = dd; a.ldk = a.ldv = 2 * dd; a.ldo = dd;
      a.q = q; a.k = io->kv[li]; a.v = (const char*)io->kv[li] + (size_t)dd * esz;
      a.klens = io->kmem; a.scale = 1.f / sqrtf((float)(dd / H));
      a.out = o; a.lse = lse;
      if (emoasr_attn_fwd(dtype, &a, stream)) return 1;
    }
    if (linear(dtype, M, dd, dd, o, dd, Ly.out2, x, EMOASR_ACT_NONE, x2, stream)) return 1;
    // feed-forward
    if (emoasr_layernorm_fwd(dtype, M, dd, x, Ly.ln3.g, Ly.ln3.b, 1e-12f, h, nullptr, nullptr, stream)) return 1;
    if (linear(dtype, M, F, dd, h, dd, Ly.w1, act, EMOASR_ACT_RELU, nullptr, stream)) return 1;
    if (linear(dtype, M, dd, F, act, F, Ly.w2, x2, EMOASR_ACT_NONE, x, stream)) return 1;
    void* t = x; x = x2; x2 = t;
  }
  if (emoasr_layernorm_fwd(dtype, M, dd, x, io->ln_out.g, io->ln_out.b, 1e-12f, h, nullptr, nullptr, stream)) return 1;
  // output projection of the last position only: rows (b, L-1) of h, row stride L*dd
  return linear(dtype, nb, io->V, dd, (const char*)h + (size_t)(L - 1) * dd * esz, (long)L * dd, io->out, io->logits_last,
                EMOASR_ACT_NONE, nullptr, stream);
}

extern "C" int emoasr_bert_lm_infer(int dtype, int nl, const emoasr_bert_layer_t* layers, const emoasr_bert_infer_t* io,
                                    void* stream) {
  EMO_CHECK(layers && io && io->ws && io->logp, "bert_lm_infer: missing arguments");
  const int nb = io->nb, L = io->L, d = io->d, H = io->H, F = io->F, V = io->V, M = nb * L;
  EMO_CHECK(nb > 0 && L > 0 && d % H == 0, "bert_lm_infer: bad dims nb=%d L=%d d=%d H=%d", nb, L, d, H);
  const size_t esz = dtype == EMO_BF16 ? 2 : 4;
  Bump ws{(char*)io->ws, io->ws_bytes};
  void* x = ws.take(M * d * esz);
  void* y = ws.take(M * d * esz);
  void* o = ws.take(M * d * esz);
  void* qkv = ws.take((size_t)M * 3 * d * esz);
  void* act = ws.take((size_t)M * F * esz);
  float* lse = (float*)ws.take((size_t)nb * H * L * 4);
  void* t1 = ws.take((size_t)nb * d * esz);
  void* t2 = ws.take((size_t)nb * d * esz);
  float* logits = (float*)ws.take((size_t)nb * V * 4);
  EMO_CHECK(ws.ok, "bert_lm_infer: scratch too small (%zu bytes given)", io->ws_bytes);
  if (emoasr_embed_fwd(dtype, M, L, d, io->ids, io->word_emb, io->pe, 1.f, 0.f, 0, y, stream)) return 1;
  if (emoasr_layernorm_fwd(dtype, M, d, y, io->ln_emb.g, io->ln_emb.b, 1e-12f, x, nullptr, nullptr, stream)) return 1;
  for (int li = 0; li < nl; ++li) {
    const emoasr_bert_layer_t& Ly = layers[li];
    if (linear(dtype, M, 3 * d, d, x, d, Ly.qkv, qkv, EMOASR_ACT_NONE, nullptr, stream)) return 1;
    if (self_attn(dtype, nb, L, d, H, qkv, io->klens, 1, o, lse, stream)) return 1;
    if (linear(dtype, M, d, d, o, d, Ly.attn_out, y, EMOASR_ACT_NONE, x, stream)) return 1;
    if (emoasr_layernorm_fwd(dtype, M, d, y, Ly.ln_attn.g, Ly.ln_attn.b, 1e-12f, x, nullptr, nullptr, stream)) return 1;
    if (linear(dtype, M, F, d, x, d, Ly.inter, act, 3 /* GELU */, nullptr, stream)) return 1;
    if (linear(dtype, M, d, F, act, F, Ly.out, y, EMOASR_ACT_NONE, x, stream)) return 1;
    if (emoasr_layernorm_fwd(dtype, M, d, y, Ly.ln_out.g, Ly.ln_out.b, 1e-12f, x, nullptr, nullptr, stream)) return 1;
  }
  // prediction head on the last position: dense + GELU, LayerNorm, tied output embedding + bias, log-softmax
  if (linear(dtype, nb, d, d, (const char*)x + (size_t)(L - 1) * d * esz, (long)L * d, io->transform, t1, 3, nullptr, stream))
    return 1;
  if (emoasr_layernorm_fwd(dtype, nb, d, t1, io->ln_transform.g, io->ln_transform.b, 1e-12f, t2, nullptr, nullptr, stream))
    return 1;
  emoasr_lin_t tied{io->word_emb, io->out_bias};
  if (linear(dtype, nb, V, d, t2, d, tied, logits, EMOASR_ACT_NONE, nullptr, stream, 1)) return 1;
  return emoasr_log_softmax(EMO_F32, nb, V, logits, V, nullptr, 0, 0.f, io->logp, V, stream);
}
