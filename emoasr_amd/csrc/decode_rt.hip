// Beam-search step runtime: the Transformer decoder (whole prefix, cached cross-attention K/V) and the
// BERT-style Transformer LM as ONE C-ABI call each per output step -- the same kernels the Python
// sequencing (engine.py:_dec_forward, modeling/lm.py:predict_device) launches one FFI call at a time,
// minus the work the beam search never uses (logits of earlier positions, K/V of the unchanged memory).
// Reference: decoders/transformer.py:148-159 + transformer.py:156-198; lm/modeling/transformer.py:62-77.
#include <math.h>
#include <stdlib.h>
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

// ---------------------------------------------------------------------------------------------------------------
// Cached single-position steps + beam bookkeeping on the device (config 4: joint CTC/attention beam search with LM
// shallow fusion, decoders/transformer.py:161-294).  The *_infer entry points above recompute the whole prefix of every
// hypothesis per output step and leave the beam bookkeeping to the host (four D2H copies and five H2D copies per step).
// Here a step touches ONE position per hypothesis: the self-attention keys / values of the earlier positions live in
// caches [layer][hypothesis][position][d] that are re-ordered by parent beam on the device, the position is read from
// device memory, and emoasr_beam_update does the reference's list bookkeeping (per-beam candidate selection, the global
// prune, <eos> handling, results) in one single-workgroup kernel -- so a whole step is a fixed sequence of launches with
// no host input: it is captured once into a HIP graph and replayed, and the host reads back one flag per step.
// ---------------------------------------------------------------------------------------------------------------
namespace {

template <typename T>
__global__ __launch_bounds__(256) void step_embed_kernel(int nb, int d, const int* __restrict__ ids, const T* __restrict__ table,
                                                         const float* __restrict__ pe, float scale, const int* __restrict__ pos,
                                                         T* __restrict__ out) {
  const int p = *pos;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nb * d; i += gridDim.x * 256) {
    const int m = i / d, c = i - m * d;
    out[i] = from_f32<T>(to_f32(table[(long)ids[m] * d + c]) * scale + pe[(long)p * d + c]);
  }
}

// kc[b][pos][:] = qkv[b][d:2d], vc[b][pos][:] = qkv[b][2d:3d]
template <typename T>
__global__ __launch_bounds__(256) void kv_append_kernel(int nb, int d, int Lmax, const T* __restrict__ qkv, T* __restrict__ kc,
                                                        T* __restrict__ vc, const int* __restrict__ pos) {
  const int p = *pos;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nb * d; i += gridDim.x * 256) {
    const int m = i / d, c = i - m * d;
    kc[((long)m * Lmax + p) * d + c] = qkv[(long)m * 3 * d + d + c];
    vc[((long)m * Lmax + p) * d + c] = qkv[(long)m * 3 * d + 2 * d + c];
  }
}

// dst[l][b][t][:] = src[l][parent[b]][t][:] for t < pos (16-byte chunks; both the K and the V cache: blockIdx.z)
__global__ __launch_bounds__(256) void cache_gather_kernel(int nl, int nb, int Lmax, int row_bytes, const char* __restrict__ sk,
                                                           const char* __restrict__ sv, char* __restrict__ dk,
                                                           char* __restrict__ dv, const int* __restrict__ parent,
                                                           const int* __restrict__ pos) {
  const int p = *pos;
  const int l = blockIdx.y / nb, b = blockIdx.y - l * nb;
  const char* src = (blockIdx.z ? sv : sk) + ((long)l * nb + parent[b]) * Lmax * row_bytes;
  char* dst = (blockIdx.z ? dv : dk) + ((long)l * nb + b) * Lmax * row_bytes;
  const int n16 = p * row_bytes / 16;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256)
    reinterpret_cast<u32x4*>(dst)[i] = reinterpret_cast<const u32x4*>(src)[i];
}

}  // namespace
// csrc/decode_coop.hip: the same in one launch of 16 cooperating workgroups with grid barriers between the stages (bf16)
bool emo_decode_coop_ok(int dtype, int nb, int nl, int max_layers, int d, int H, int F, int T);
int emo_bert_lm_step_coop(int nl, const emoasr_bert_layer_t* layers, const emoasr_bert_step_t* io, void* out_hidden, hipStream_t s);
int emo_transformer_decoder_step_coop(int nl, const emoasr_decoder_layer_t* layers, const emoasr_decoder_step_t* io, void* out_x,
                                      hipStream_t s);
namespace {

// y = act(LN?(x) . W^T + b) (+ r | LN(r)) for the step's <= 16 rows (bf16)
int rl(int M, int N, int K, const void* x, const emoasr_lnp_t* lna, const emoasr_lin_t& l, int act, const void* res,
       const emoasr_lnp_t* lnr, void* y, int out_f32, void* stream) {
  return emoasr_rowlin(M, N, K, x, K, lna ? lna->g : nullptr, lna ? lna->b : nullptr, 1e-12f, l.w, l.b, act, res, N,
                       lnr ? lnr->g : nullptr, lnr ? lnr->b : nullptr, 1e-12f, y, out_f32, N, stream);
}

int attn_cached(int dtype, int nb, int Lmax, int d, int H, const void* qkv, const void* kc, const void* vc, const int* klens,
                void* o, float* lse, void* stream) {
  emoasr_attn_t a{};
  a.B = nb; a.H = H; a.DK = d / H; a.Tq = 1; a.Tk = Lmax;
  a.ldq = 3 * d; a.ldk = a.ldv = d; a.ldo = d;
  a.q = qkv; a.k = kc; a.v = vc;
  a.klens = klens; a.causal = 0; a.scale = 1.f / sqrtf((float)(d / H));
  a.out = o; a.lse = lse;
  return emoasr_attn_fwd(dtype, &a, stream);
}

}  // namespace

extern "C" size_t emoasr_decode_step_ws_bytes(int dtype, int nb, int d, int H, int F, int V) {
  const size_t esz = dtype == EMO_BF16 ? 2 : 4;
  return (size_t)nb * esz * ((size_t)5 * d + 3 * d + F + 2 * d) + (size_t)nb * H * 4 + (size_t)nb * V * 4 + 20 * 256;
}

extern "C" int emoasr_beam_cache_gather(int dtype, int nl, int nb, int Lmax, int d, const void* src_k, const void* src_v,
                                        void* dst_k, void* dst_v, const int* parent, const int* pos, void* stream) {
  const int row_bytes = d * (dtype == EMO_BF16 ? 2 : 4);
  EMO_CHECK(row_bytes % 16 == 0, "beam_cache_gather: rows must be multiples of 16 bytes");
  dim3 grid(4, nl * nb, 2);
  cache_gather_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(nl, nb, Lmax, row_bytes, (const char*)src_k, (const char*)src_v,
                                                             (char*)dst_k, (char*)dst_v, parent, pos);
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_transformer_decoder_step(int dtype, int nl, const emoasr_decoder_layer_t* layers,
                                               const emoasr_decoder_step_t* io, void* stream) {
  EMO_CHECK(layers && io && io->ws && io->kv && io->logits_last && io->kcache && io->vcache && io->pos && io->klens,
            "decoder_step: missing arguments");
  const int nb = io->nb, Lmax = io->Lmax, T = io->T, dd = io->dd, H = io->H, F = io->F;
  EMO_CHECK(nb > 0 && Lmax > 0 && T > 0 && dd % H == 0, "decoder_step: bad dims nb=%d Lmax=%d T=%d dd=%d H=%d", nb, Lmax, T, dd, H);
  const size_t esz = dtype == EMO_BF16 ? 2 : 4;
  hipStream_t s = (hipStream_t)stream;
  Bump ws{(char*)io->ws, io->ws_bytes};
  void* x = ws.take(nb * dd * esz);
  void* x2 = ws.take(nb * dd * esz);
  void* h = ws.take(nb * dd * esz);
  void* o = ws.take(nb * dd * esz);
  void* q = ws.take(nb * dd * esz);
  void* qkv = ws.take((size_t)nb * 3 * dd * esz);
  void* act = ws.take((size_t)nb * F * esz);
  float* lse = (float*)ws.take((size_t)nb * H * 4);
  EMO_CHECK(ws.ok, "decoder_step: scratch too small (%zu bytes given)", io->ws_bytes);
  if (emo_decode_coop_ok(dtype, nb, nl, 8, dd, H, F, T)) {
    // the whole stack in one launch of 16 cooperating workgroups, then the final LayerNorm + vocabulary projection across the chip
    if (emo_transformer_decoder_step_coop(nl, layers, io, x, s)) return 1;
    return rl(nb, io->V, dd, x, &io->ln_out, io->out, EMOASR_ACT_NONE, nullptr, nullptr, io->logits_last, 0, stream);
  }
  EMO_DISPATCH(dtype, (step_embed_kernel<T><<<cdiv(nb * dd, 256), 256, 0, s>>>(nb, dd, io->ids, (const T*)io->embed, io->pe,
                                                                                io->emb_scale, io->pos, (T*)x)));
  const size_t layer_bytes = (size_t)nb * Lmax * dd * esz;
  for (int li = 0; li < nl; ++li) {
    const emoasr_decoder_layer_t& Ly = layers[li];
    char* kc = (char*)io->kcache + li * layer_bytes;
    char* vc = (char*)io->vcache + li * layer_bytes;
    if (emoasr_layernorm_fwd(dtype, nb, dd, x, Ly.ln1.g, Ly.ln1.b, 1e-12f, h, nullptr, nullptr, stream)) return 1;
    if (linear(dtype, nb, 3 * dd, dd, h, dd, Ly.qkv, qkv, EMOASR_ACT_NONE, nullptr, stream)) return 1;
    EMO_DISPATCH(dtype, (kv_append_kernel<T><<<cdiv(nb * dd, 256), 256, 0, s>>>(nb, dd, Lmax, (const T*)qkv, (T*)kc, (T*)vc, io->pos)));
    if (attn_cached(dtype, nb, Lmax, dd, H, qkv, kc, vc, io->klens, o, lse, stream)) return 1;
    if (linear(dtype, nb, dd, dd, o, dd, Ly.out, x2, EMOASR_ACT_NONE, x, stream)) return 1;
    if (emoasr_layernorm_fwd(dtype, nb, dd, x2, Ly.ln2.g, Ly.ln2.b, 1e-12f, h, nullptr, nullptr, stream)) return 1;
    if (linear(dtype, nb, dd, dd, h, dd, Ly.q2, q, EMOASR_ACT_NONE, nullptr, stream)) return 1;
    {
      emoasr_attn_t a{};
      a.B = nb; a.H = H; a.DK = dd / H; a.Tq = 1; a.Tk = T;
      a.ldq = dd; a.ldk = a.ldv = 2 * dd; a.ldo = dd;
      a.q = q; a.k = io->kv[li]; a.v = (const char*)io->kv[li] + (size_t)dd * esz;
      a.klens = io->kmem; a.scale = 1.f / sqrtf((float)(dd / H));
      a.out = o; a.lse = lse;
      if (emoasr_attn_fwd(dtype, &a, stream)) return 1;
    }
    if (linear(dtype, nb, dd, dd, o, dd, Ly.out2, x, EMOASR_ACT_NONE, x2, stream)) return 1;
    if (emoasr_layernorm_fwd(dtype, nb, dd, x, Ly.ln3.g, Ly.ln3.b, 1e-12f, h, nullptr, nullptr, stream)) return 1;
    if (linear(dtype, nb, F, dd, h, dd, Ly.w1, act, EMOASR_ACT_RELU, nullptr, stream)) return 1;
    if (linear(dtype, nb, dd, F, act, F, Ly.w2, x2, EMOASR_ACT_NONE, x, stream)) return 1;
    void* t = x; x = x2; x2 = t;
  }
  if (emoasr_layernorm_fwd(dtype, nb, dd, x, io->ln_out.g, io->ln_out.b, 1e-12f, h, nullptr, nullptr, stream)) return 1;
  EMO_LAUNCH_CHECK();
  return linear(dtype, nb, io->V, dd, h, dd, io->out, io->logits_last, EMOASR_ACT_NONE, nullptr, stream);
}

extern "C" int emoasr_bert_lm_step(int dtype, int nl, const emoasr_bert_layer_t* layers, const emoasr_bert_step_t* io,
                                   void* stream) {
  EMO_CHECK(layers && io && io->ws && io->logp && io->kcache && io->vcache && io->pos && io->klens, "bert_lm_step: missing arguments");
  const int nb = io->nb, Lmax = io->Lmax, d = io->d, H = io->H, F = io->F, V = io->V;
  EMO_CHECK(nb > 0 && Lmax > 0 && d % H == 0, "bert_lm_step: bad dims nb=%d Lmax=%d d=%d H=%d", nb, Lmax, d, H);
  const size_t esz = dtype == EMO_BF16 ? 2 : 4;
  hipStream_t s = (hipStream_t)stream;
  Bump ws{(char*)io->ws, io->ws_bytes};
  void* x = ws.take(nb * d * esz);
  void* y = ws.take(nb * d * esz);
  void* o = ws.take(nb * d * esz);
  void* qkv = ws.take((size_t)nb * 3 * d * esz);
  void* act = ws.take((size_t)nb * F * esz);
  float* lse = (float*)ws.take((size_t)nb * H * 4);
  void* t1 = ws.take((size_t)nb * d * esz);
  void* t2 = ws.take((size_t)nb * d * esz);
  float* logits = (float*)ws.take((size_t)nb * V * 4);
  EMO_CHECK(ws.ok, "bert_lm_step: scratch too small (%zu bytes given)", io->ws_bytes);
  if (emo_decode_coop_ok(dtype, nb, nl, 12, d, H, F, 0)) {
    if (emo_bert_lm_step_coop(nl, layers, io, t1, s)) return 1;
    emoasr_lin_t tied_co{io->word_emb, io->out_bias};
    if (io->raw_logits) return rl(nb, V, d, t1, &io->ln_transform, tied_co, EMOASR_ACT_NONE, nullptr, nullptr, io->logp, 1, stream);
    if (rl(nb, V, d, t1, &io->ln_transform, tied_co, EMOASR_ACT_NONE, nullptr, nullptr, logits, 1, stream)) return 1;
    return emoasr_log_softmax(EMO_F32, nb, V, logits, V, nullptr, 0, 0.f, io->logp, V, stream);
  }
  EMO_DISPATCH(dtype, (step_embed_kernel<T><<<cdiv(nb * d, 256), 256, 0, s>>>(nb, d, io->ids, (const T*)io->word_emb, io->pe, 1.f,
                                                                              io->pos, (T*)y)));
  const size_t layer_bytes = (size_t)nb * Lmax * d * esz;
  if (emoasr_layernorm_fwd(dtype, nb, d, y, io->ln_emb.g, io->ln_emb.b, 1e-12f, x, nullptr, nullptr, stream)) return 1;
  for (int li = 0; li < nl; ++li) {
    const emoasr_bert_layer_t& Ly = layers[li];
    char* kc = (char*)io->kcache + li * layer_bytes;
    char* vc = (char*)io->vcache + li * layer_bytes;
    if (linear(dtype, nb, 3 * d, d, x, d, Ly.qkv, qkv, EMOASR_ACT_NONE, nullptr, stream)) return 1;
    EMO_DISPATCH(dtype, (kv_append_kernel<T><<<cdiv(nb * d, 256), 256, 0, s>>>(nb, d, Lmax, (const T*)qkv, (T*)kc, (T*)vc, io->pos)));
    if (attn_cached(dtype, nb, Lmax, d, H, qkv, kc, vc, io->klens, o, lse, stream)) return 1;
    if (linear(dtype, nb, d, d, o, d, Ly.attn_out, y, EMOASR_ACT_NONE, x, stream)) return 1;
    if (emoasr_layernorm_fwd(dtype, nb, d, y, Ly.ln_attn.g, Ly.ln_attn.b, 1e-12f, x, nullptr, nullptr, stream)) return 1;
    if (linear(dtype, nb, F, d, x, d, Ly.inter, act, 3 /* GELU */, nullptr, stream)) return 1;
    if (linear(dtype, nb, d, F, act, F, Ly.out, y, EMOASR_ACT_NONE, x, stream)) return 1;
    if (emoasr_layernorm_fwd(dtype, nb, d, y, Ly.ln_out.g, Ly.ln_out.b, 1e-12f, x, nullptr, nullptr, stream)) return 1;
  }
  if (linear(dtype, nb, d, d, x, d, io->transform, t1, 3, nullptr, stream)) return 1;
  if (emoasr_layernorm_fwd(dtype, nb, d, t1, io->ln_transform.g, io->ln_transform.b, 1e-12f, t2, nullptr, nullptr, stream))
    return 1;
  emoasr_lin_t tied{io->word_emb, io->out_bias};
  if (io->raw_logits) return linear(dtype, nb, V, d, t2, d, tied, io->logp, EMOASR_ACT_NONE, nullptr, stream, 1);
  if (linear(dtype, nb, V, d, t2, d, tied, logits, EMOASR_ACT_NONE, nullptr, stream, 1)) return 1;
  EMO_LAUNCH_CHECK();
  return emoasr_log_softmax(EMO_F32, nb, V, logits, V, nullptr, 0, 0.f, io->logp, V, stream);
}

// ---- beam bookkeeping (decoders/transformer.py:215-290, restated in emoasr_amd/modeling/beam_search.py) ------------
namespace {

// One workgroup.  cw_eff candidates per live beam m < n_alive:
//   sc[m][j]  = f32( f32(1 - lam) * vals + f32(lam) * (psi - score_ctc[m]) ) (+ f32(mu) * lm_at)       numpy float32 semantics
//   per beam: the bw best j (stable: ties to the lower j);  candidate score = score[m] + (double)sc    Python float
//   all candidates: the bw best (stable in (m, rank) order);  then, in that order: <eos> -> result (dropped when the
//   hypothesis is empty), else alive.  psi == NULL: no CTC term (vals are the top-bw attention(+LM) scores).
__global__ __launch_bounds__(256) void beam_update_kernel(emoasr_beam_update_t u) {
  __shared__ float sc[32][32];
  __shared__ int sel_j[32][32];        // per beam: candidate index by rank
  __shared__ double cscore[1024];
  __shared__ int c_m[1024], c_j[1024], c_rank[1024];
  __shared__ int sorted_e[32];
  emoasr_beam_state_t* st = u.state;
  if (st->done) return;
  const int tid = threadIdx.x;
  const int bw = u.bw, cw = u.cw, na = st->n_alive, pos = st->pos;
  const float f1 = u.one_minus_lam, fl = u.lam, fm = u.mu;
  for (int i = tid; i < na * cw; i += 256) {
    const int m = i / cw, j = i - m * cw;
    float s;
    if (u.psi) {
      s = f1 * u.vals[m * cw + j] + fl * (u.psi[m * cw + j] - u.score_ctc[m]);
      if (u.lm_at) s = s + fm * u.lm_at[m * cw + j];
    } else {
      s = u.vals[m * cw + j];
    }
    sc[m][j] = s;
  }
  __syncthreads();
  const int keep = min(bw, cw);
  for (int i = tid; i < na * cw; i += 256) {
    const int m = i / cw, j = i - m * cw;
    const float s = sc[m][j];
    int r = 0;
    for (int k = 0; k < cw; ++k) r += (sc[m][k] > s) || (sc[m][k] == s && k < j);
    if (r < keep) sel_j[m][r] = j;
  }
  __syncthreads();
  const int nc = na * keep;
  for (int e = tid; e < nc; e += 256) {
    const int m = e / keep, r = e - m * keep, j = sel_j[m][r];
    c_m[e] = m; c_j[e] = j;
    cscore[e] = u.score[m] + (double)sc[m][j];
  }
  __syncthreads();
  for (int e = tid; e < nc; e += 256) {
    const double s = cscore[e];
    int r = 0;
    for (int k = 0; k < nc; ++k) r += (cscore[k] > s) || (cscore[k] == s && k < e);
    c_rank[e] = r;
    if (r < bw) sorted_e[r] = e;
  }
  __syncthreads();
  if (tid != 0) return;
  int a = 0, nres = st->n_results;
  const int nsel = min(bw, nc);
  int* hp = u.hist_parent + (long)pos * bw;
  int* ht = u.hist_token + (long)pos * bw;
  for (int r = 0; r < nsel; ++r) {
    const int e = sorted_e[r], m = c_m[e], j = c_j[e];
    const int tok = u.cands[m * cw + j];
    if (tok == u.eos) {
      if (pos < 1) continue;  // empty hypothesis
      u.res_score[nres] = cscore[e] + u.len_weight * (double)(pos + 2);
      u.res_step[nres] = pos;
      u.res_parent[nres] = m;
      ++nres;
      if (nres >= bw) break;
    } else {
      u.n_ids[a] = tok; u.n_parent[a] = m; u.n_pcand[a] = j;
      u.score[a] = cscore[e];
      u.score_ctc[a] = u.psi ? u.psi[m * cw + j] : 0.f;
      hp[a] = m; ht[a] = tok;
      ++a;
    }
  }
  for (int k = a; k < bw; ++k) {  // unused slots: copies of slot 0, so that the networks run on valid ids / caches
    u.n_ids[k] = a > 0 ? u.n_ids[0] : u.eos;
    u.n_parent[k] = a > 0 ? u.n_parent[0] : 0;
    u.n_pcand[k] = a > 0 ? u.n_pcand[0] : 0;
    u.score[k] = 0.0; u.score_ctc[k] = 0.f;
    hp[k] = -1; ht[k] = -1;
  }
  for (int k = 0; k < bw; ++k) {
    u.n_last[k] = u.n_ids[k];
    u.n_outlen[k] = pos + 1;
    u.n_klens[k] = pos + 2;
  }
  st->n_alive = a;
  st->n_results = nres;
  st->done = (nres >= bw || a == 0) ? 1 : 0;
  st->pos = pos + 1;
  if (u.host_mirror) {   // the host's view of the search: written last, `pos` after the rest
    volatile int* hm = u.host_mirror;
    hm[1] = a; hm[2] = nres; hm[3] = st->done;
    __threadfence_system();
    hm[0] = pos + 1;
    __threadfence_system();
  }
}

}  // namespace

extern "C" int emoasr_beam_update(const emoasr_beam_update_t* u, void* stream) {
  EMO_CHECK(u && u->state && u->vals && u->cands && u->score && u->n_ids, "beam_update: missing arguments");
  EMO_CHECK(u->bw >= 1 && u->bw <= 32 && u->cw >= 1 && u->cw <= 32, "beam_update: beam %d / candidates %d outside 1..32", u->bw, u->cw);
  beam_update_kernel<<<1, 256, 0, (hipStream_t)stream>>>(*u);
  EMO_LAUNCH_CHECK();
  return 0;
}

// ---- one whole output step of the joint search: every launch of it, from one C-ABI call ------------------------------
// main stream:  cache gather (both networks) -> decoder step ---------------------------\
// side stream:                               \-> LM step (independent chain) ------------+-> log-softmax + LM fusion -> top-cw
//                                                                                            -> CTC prefix scores -> beam update
// side == NULL (or no LM): everything on the main stream.
// `parts`: 1 = decoder chain (its cache gather + the decoder step), 2 = LM chain (gather + LM step), 4 = scoring tail.  The two
// chains are independent; the tail needs both.  emoasr_joint_beam_step runs all three with a fork / join over two streams;
// the graph path captures one graph per part, so that the chains replay concurrently on two streams (a single captured
// graph replayed its branches one after the other: 0.83 ms per step for ~185 kernels).
extern "C" int emoasr_joint_beam_step_parts(int dtype, const emoasr_joint_step_t* js, int parts, void* stream) {
  EMO_CHECK(js && js->dec_layers, "joint_beam_step: missing arguments");
  hipStream_t s = (hipStream_t)stream;
  const emoasr_decoder_step_t& d = js->dec;
  const emoasr_bert_step_t& l = js->lm;
  const int bw = d.nb, V = d.V;
  const bool use_lm = js->lm_nl > 0;
  if (parts & 1) {
    if (emoasr_beam_cache_gather(dtype, js->dec_nl, bw, d.Lmax, d.dd, js->dec_k_prev, js->dec_v_prev, d.kcache, d.vcache,
                                 js->parent, d.pos, s)) return 1;
    if (emoasr_transformer_decoder_step(dtype, js->dec_nl, js->dec_layers, &d, s)) return 1;
  }
  if ((parts & 2) && use_lm) {
    if (emoasr_beam_cache_gather(dtype, js->lm_nl, bw, l.Lmax, l.d, js->lm_k_prev, js->lm_v_prev, l.kcache, l.vcache, js->parent,
                                 l.pos, s)) return 1;
    if (emoasr_bert_lm_step(dtype, js->lm_nl, js->lm_layers, &l, s)) return 1;
  }
  if (parts & 4) {
    // scores_att (+ mu * lm: the reference adds the LM term in place, so it is inside scores_att when the CTC re-scoring
    // adds it again -- see modeling/beam_search.py)
    const emoasr_beam_update_t& u = js->upd;
    if (use_lm && l.raw_logits) {
      // one launch: both log-softmaxes, the fusion and the candidate selection (the LM step left its raw logits in l.logp)
      if (emoasr_beam_scores_topk(dtype, bw, V, u.cw, d.logits_last, V, l.logp, V, js->upd.mu, (float*)u.vals, (int*)u.cands,
                                  (float*)u.lm_at, s)) return 1;
    } else if (!use_lm) {
      if (emoasr_beam_scores_topk(dtype, bw, V, u.cw, d.logits_last, V, nullptr, 0, 0.f, (float*)u.vals, (int*)u.cands, nullptr,
                                  s)) return 1;
    } else {
      if (emoasr_log_softmax(dtype, bw, V, d.logits_last, V, l.logp, V, js->upd.mu, js->scores_pre, V, s)) return 1;
      const bool want_aux = u.lm_at != nullptr;
      if (emoasr_topk(bw, V, u.cw, js->scores_pre, V, want_aux ? l.logp : nullptr, V, (float*)u.vals, (int*)u.cands,
                      want_aux ? (float*)u.lm_at : nullptr, s)) return 1;
    }
    if (u.psi) {
      if (emoasr_ctc_prefix_score(bw, js->T, V, u.cw, js->ctc_x, js->states_prev, u.cw, js->parent, u.n_pcand, nullptr, u.n_last,
                                  u.n_outlen, u.cands, js->blank, u.eos, (float*)u.psi, js->states_cur, s)) return 1;
    }
    if (emoasr_beam_update(&u, s)) return 1;
  }
  return 0;
}

extern "C" int emoasr_joint_beam_step(int dtype, const emoasr_joint_step_t* js, void* stream, void* side_stream) {
  EMO_CHECK(js && js->dec_layers, "joint_beam_step: missing arguments");
  hipStream_t s = (hipStream_t)stream, s2 = (hipStream_t)side_stream;
  static hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  if (!ev_fork) {
    hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming);
    hipEventCreateWithFlags(&ev_join, hipEventDisableTiming);
  }
  const bool fork = js->lm_nl > 0 && s2 != nullptr && s2 != s;
  if (fork) {
    hipEventRecord(ev_fork, s);
    hipStreamWaitEvent(s2, ev_fork, 0);
    if (emoasr_joint_beam_step_parts(dtype, js, 2, s2)) return 1;
    hipEventRecord(ev_join, s2);
    if (emoasr_joint_beam_step_parts(dtype, js, 1, s)) return 1;
    hipStreamWaitEvent(s, ev_join, 0);
    return emoasr_joint_beam_step_parts(dtype, js, 4, s);
  }
  return emoasr_joint_beam_step_parts(dtype, js, 7, s);
}

// ---- the step as a HIP graph ---------------------------------------------------------------------------------------------
// A step is ~190 launches of 3-5 us kernels; issued one by one they cost the host ~5.5 us each (1.1 ms per step, twice the
// GPU time).  emoasr_joint_beam_graph_build captures emoasr_joint_beam_step (both streams) into a graph for `slot` (0 / 1:
// the even / odd steps differ in which cache and scorer-state buffers are "previous" and "current"); for the next utterance
// the same topology is re-captured with the new pointers and the instantiated graph is updated in place
// (hipGraphExecUpdate), which is far cheaper than a new instantiation.  emoasr_joint_beam_graph_launch replays it.
// (One search at a time per process -- the reference decodes one utterance at a time, decoders/transformer.py:181 -- so the
// instantiated graphs live in six process-wide slots.)
namespace {
hipGraphExec_t g_beam_exec[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // [slot][part: decoder, LM, tail]
}

// part: 0 = decoder chain, 1 = LM chain, 2 = scoring tail (emoasr_joint_beam_step_parts masks 1 / 2 / 4)
extern "C" int emoasr_joint_beam_graph_build(int dtype, const emoasr_joint_step_t* js, int slot, int part, void* stream) {
  EMO_CHECK((slot == 0 || slot == 1) && part >= 0 && part < 3, "joint_beam_graph_build: slot %d part %d", slot, part);
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
  if (e != hipSuccess) { emo_set_error("hipStreamBeginCapture: %s", hipGetErrorString(e)); return 1; }
  const int rc = emoasr_joint_beam_step_parts(dtype, js, 1 << part, stream);
  hipGraph_t graph = nullptr;
  e = hipStreamEndCapture(s, &graph);
  if (rc) { if (graph) hipGraphDestroy(graph); return rc; }
  if (e != hipSuccess || !graph) { emo_set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return 1; }
  hipGraphExec_t& ex = g_beam_exec[slot * 3 + part];
  bool ok = false;
  static const bool no_update = getenv("EMOASR_GRAPH_NO_UPDATE") != nullptr;
  if (ex && no_update) { hipGraphExecDestroy(ex); ex = nullptr; }
  if (ex) {
    hipGraphNode_t bad = nullptr;
    hipGraphExecUpdateResult res;
    ok = hipGraphExecUpdate(ex, graph, &bad, &res) == hipSuccess;
    if (!ok) {
      (void)hipGetLastError();
      hipGraphExecDestroy(ex);
      ex = nullptr;
    }
  }
  if (!ok) {
    e = hipGraphInstantiate(&ex, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { hipGraphDestroy(graph); emo_set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); return 1; }
  }
  hipGraphDestroy(graph);
  return 0;
}

extern "C" int emoasr_joint_beam_graph_launch(int slot, int part, void* stream) {
  EMO_CHECK((slot == 0 || slot == 1) && part >= 0 && part < 3 && g_beam_exec[slot * 3 + part],
            "joint_beam_graph_launch: slot %d part %d has no graph", slot, part);
  hipError_t e = hipGraphLaunch(g_beam_exec[slot * 3 + part], (hipStream_t)stream);
  if (e != hipSuccess) { emo_set_error("hipGraphLaunch: %s", hipGetErrorString(e)); return 1; }
  return 0;
}
