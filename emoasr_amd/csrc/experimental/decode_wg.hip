// EXPERIMENT, off by default -- the cached decode steps of config 4 as ONE workgroup per network (bf16): the whole
// Transformer-LM stack, or the whole Transformer decoder stack, for the <= 16 live hypotheses of a beam-search step in a
// single launch.
//
// Why it was tried: at beam 10 a step of either network is ~90 dependent launches of 3-6 us each (csrc/decode_rt.hip), i.e.
// the step time is the launch chain (0.68 ms), while the work itself is one pass over the weights (19 MB LM, 9 MB decoder)
// for 10 rows.  One workgroup of 1024 threads streams memory at ~106 GB/s (measured, tools/micro/one_wg_bw.hip): 19 MB in
// ~180 us with NO launch boundaries and no cross-workgroup synchronisation -- every activation of the step (10 rows x
// <= 1024 values) lives in LDS from the embedding to the last layer; only the K / V caches, the encoder memory and the
// weights are read from global memory.  The vocabulary projections (5 MB each) stay separate multi-workgroup launches
// (emoasr_rowlin with the final LayerNorm folded in).
//
// MEASURED (MI355X, beam 10, d 256, F 1024, rocprofv3): LM stack 780 us at position 0 .. 1000 us at position 35 per step,
// decoder stack 1.2 - 2.0 ms (T' 224 - 452): 4-7x the streaming estimate and slower than the launch chain it was meant to
// replace (step 1.5 - 2.0 ms against 0.68 ms).  Each of the 84 (LM) / 78 (decoder) barrier-separated stages pays its own
// global-memory round trip (~2 us on one CU at low occupancy) before its first MFMA -- software-pipelining the weight
// fragments INSIDE a stage changed nothing (872 vs 883 us) -- and the cross-attention over T' keys for 40 (hypothesis, head)
// pairs on one CU costs another 100-250 us per layer.  A version that wins would have to prefetch the NEXT stage's weights
// across the barrier (an LDS weight ring) and leave the cross-attention to a chip-wide launch; not built.  The kernels are
// correct (tests/test_l3_gpu.py runs the search through them) and stay behind emoasr_set_option("decode_wg", 1).
//
//   linear layers : 16 waves x 16-column strips of 16x16x32 MFMAs; weight fragments straight from global memory
//                   (16 B per lane, 8 in flight per wave), activation fragments from LDS
//   LayerNorm     : one wave per row
//   attention     : one (hypothesis, head) pair per wave at a time, 64 keys per pass (a lane scores one key), online
//                   softmax, then a lane per output dimension; the new key / value are appended to the caches first
//
// Reference: decoders/transformer.py:148-159 + transformer.py:156-198 (pre-LN decoder layer, ReLU);
// lm/modeling/transformer.py:62-77 over modeling_bert.py:159-303,360-436 (post-LN block, GELU).
#include <math.h>
#include "../common.h"
#include "../../../include/emoasr_hip.h"

namespace {

constexpr int WG_ROWS = 16;
constexpr int WG_MAXD = 1024;   // widest activation row (feed-forward width)
constexpr int WG_THREADS = 1024;
constexpr int WG_MAXDM = 512;   // widest model dimension (LayerNorm rows): 8 values per lane

struct WgLayerLm { emoasr_bert_layer_t p; };
struct LmWgArgs {
  int nl, nb, Lmax, d, H, F;
  const int* ids; const int* pos;
  const bf16* word_emb; const float* pe; emoasr_lnp_t ln_emb;
  bf16* kcache; bf16* vcache;
  emoasr_lin_t transform;
  bf16* out_hidden;            // [nb, d]: GELU(transform(x)) -- the LM head's LayerNorm + tied projection follow in emoasr_rowlin
  emoasr_bert_layer_t layers[12];
};
struct DecWgArgs {
  int nl, nb, Lmax, T, d, H, F;
  const int* ids; const int* pos;
  const bf16* embed; const float* pe; float emb_scale;
  bf16* kcache; bf16* vcache;
  const void* kv[8];           // cross-attention K | V of the encoder memory per layer: [nb][T][2d]
  const int* kmem;
  bf16* out_x;                 // [nb, d]: the stack's output before the final LayerNorm
  emoasr_decoder_layer_t layers[8];
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// dst[r][0..d) = LayerNorm(src[r][0..d)) for the 16 rows, one wave per row (eps 1e-12, as both networks use)
__device__ void wg_layernorm(const bf16* src, int lds, bf16* dst, int ldd, int d, const emoasr_lnp_t& ln, int wave, int lane) {
  const int r = wave;
  float v[WG_MAXDM / 64], gg[WG_MAXDM / 64], bb[WG_MAXDM / 64];
  const __amdgpu_buffer_rsrc_t rsg = make_rsrc(ln.g), rsb = make_rsrc(ln.b);
#pragma unroll
  for (int i = 0; i < WG_MAXDM / 64; ++i) {   // gamma / beta on their way while the statistics are reduced
    const int k = lane + 64 * i;
    gg[i] = buf_load_f32<float>(rsg, k < d ? (unsigned)(k * 4) : EMO_OOB);
    bb[i] = buf_load_f32<float>(rsb, k < d ? (unsigned)(k * 4) : EMO_OOB);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < WG_MAXDM / 64; ++i) {
    const int k = lane + 64 * i;
    v[i] = k < d ? (float)src[r * lds + k] : 0.f;
    s += v[i];
  }
  const float mean = wave_sum(s) / d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < WG_MAXDM / 64; ++i) {
    const float dl = lane + 64 * i < d ? v[i] - mean : 0.f;
    q += dl * dl;
  }
  const float rstd = rsqrtf(wave_sum(q) / d + 1e-12f);
#pragma unroll
  for (int i = 0; i < WG_MAXDM / 64; ++i) {
    const int k = lane + 64 * i;
    if (k < d) dst[r * ldd + k] = (bf16)((v[i] - mean) * rstd * gg[i] + bb[i]);
  }
}

// For every (row m < 16, column n < N): epi(m, n, bias[n] + sum_k xs[m][k] * W[n][k]).  Strips of 16 columns are dealt to
// the 16 waves; a wave's (strip, 128-deep k chunk) iterations are software-pipelined: the 4 weight-fragment loads (and the
// bias) of the next iteration are in flight while the current one is multiplied, so the wave waits for memory once, not once
// per chunk.
template <typename Epi>
__device__ void wg_linear(const bf16* xs, int ldx, int K, const bf16* W, const float* bias, int N, int wave, int lane, Epi epi) {
  const __amdgpu_buffer_rsrc_t rsw = make_rsrc(W), rsb = make_rsrc(bias);
  const bf16* xrow = xs + (lane & 15) * ldx + 8 * (lane >> 4);
  constexpr int CK = 128, NF = CK / 32;   // k extent and fragments of one pipeline stage
  const int nchunk = (K + CK - 1) / CK;
  const int nstrip = (N + 15) / 16;
  const int my_strips = nstrip > wave ? (nstrip - wave + 15) / 16 : 0;
  const int iters = my_strips * nchunk;
  if (iters == 0) return;
  auto issue = [&](int it, bf16x8 (&wf)[NF], float& bv) {
    const int strip = wave + 16 * (it / nchunk), kc = CK * (it % nchunk);
    const int col = strip * 16 + (lane & 15);
    const bool cok = col < N;
    const unsigned woff = (unsigned)(((long)(cok ? col : 0) * K + 8 * (lane >> 4)) * 2);
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const int k0 = kc + 32 * i;
      wf[i] = buf_load16<bf16>(rsw, (cok && k0 < K) ? woff + (unsigned)(k0 * 2) : EMO_OOB).v;
    }
    if (kc == 0) bv = buf_load_f32<float>(rsb, cok ? (unsigned)(col * 4) : EMO_OOB);
  };
  bf16x8 wa[NF], wb[NF];
  float ba = 0.f, bb = 0.f, bcur = 0.f;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  issue(0, wa, ba);
  for (int it = 0; it < iters; it += 2) {
    // ---- even iteration: multiply wa, prefetch wb ----
    if (it + 1 < iters) issue(it + 1, wb, bb);
    {
      const int kc = CK * (it % nchunk);
      if (kc == 0) { acc = f32x4{0.f, 0.f, 0.f, 0.f}; bcur = ba; }
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int k0 = kc + 32 * i;
        if (k0 < K) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(xrow + k0), wa[i], acc, 0, 0, 0);
      }
      if (it % nchunk == nchunk - 1) {
        const int col = (wave + 16 * (it / nchunk)) * 16 + (lane & 15);
        if (col < N) {
#pragma unroll
          for (int r = 0; r < 4; ++r) epi(4 * (lane >> 4) + r, col, acc[r] + bcur);
        }
      }
    }
    if (it + 1 >= iters) break;
    // ---- odd iteration: multiply wb, prefetch wa ----
    if (it + 2 < iters) issue(it + 2, wa, ba);
    {
      const int it1 = it + 1, kc = CK * (it1 % nchunk);
      if (kc == 0) { acc = f32x4{0.f, 0.f, 0.f, 0.f}; bcur = bb; }
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int k0 = kc + 32 * i;
        if (k0 < K) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(xrow + k0), wb[i], acc, 0, 0, 0);
      }
      if (it1 % nchunk == nchunk - 1) {
        const int col = (wave + 16 * (it1 / nchunk)) * 16 + (lane & 15);
        if (col < N) {
#pragma unroll
          for (int r = 0; r < 4; ++r) epi(4 * (lane >> 4) + r, col, acc[r] + bcur);
        }
      }
    }
  }
}

// One (hypothesis, head) pair on one wave: out[0..dk) = softmax(q . K^T * scale) . V over `nkeys` keys whose rows are
// kbase + t * kstride / vbase + t * vstride (bf16, dk <= 128 contiguous values).  q: dk floats in LDS (wave-private),
// prob: 64 floats of wave-private LDS.  Loads are issued in groups before they are used (a dependent load per multiply
// would cost a memory round trip each).
__device__ void wg_attend(const float* q, int dk, const bf16* kbase, long kstride, const bf16* vbase, long vstride, int nkeys,
                          float scale, float* prob, bf16* out, int lane) {
  const __amdgpu_buffer_rsrc_t rsk = make_rsrc(kbase), rsv = make_rsrc(vbase);
  float m_run = -INFINITY, l_run = 0.f;
  float o0 = 0.f, o1 = 0.f;   // this lane's output dimensions: lane and lane + 64
  const int nch = dk / 8;     // 16-byte chunks per key row (<= 16)
  for (int t0 = 0; t0 < nkeys; t0 += 64) {
    const int t = t0 + lane;
    const bool tok = t < nkeys;
    float s = 0.f;
#pragma unroll
    for (int c0 = 0; c0 < 16; c0 += 8) {   // 8 chunks (32 registers) at a time: the kernel runs at 128 VGPRs per thread
      if (c0 < nch) {
        bf16x8 kv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c)
          kv[c] = buf_load16<bf16>(rsk, (tok && c0 + c < nch) ? (unsigned)(((long)t * kstride + 8 * (c0 + c)) * 2) : EMO_OOB).v;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          if (c0 + c < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) s += q[8 * (c0 + c) + e] * (float)kv[c][e];
          }
        }
      }
    }
    s = tok ? s * scale : -INFINITY;
    const float m_new = fmaxf(m_run, wave_max(s));
    const float p = tok ? __expf(s - m_new) : 0.f;
    const float corr = __expf(m_run - m_new);   // 0 on the first pass (m_run = -inf)
    l_run = l_run * corr + wave_sum(p);
    o0 *= corr; o1 *= corr;
    m_run = m_new;
    prob[lane] = p;   // (wave-private LDS: visible to this wave's reads below in program order)
    const int n = min(64, nkeys - t0);
    for (int j0 = 0; j0 < n; j0 += 8) {
      float v0[8], v1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool ok = j0 + j < n;
        const long row = (long)(t0 + j0 + j) * vstride;
        v0[j] = buf_load_f32<bf16>(rsv, (ok && lane < dk) ? (unsigned)((row + lane) * 2) : EMO_OOB);
        v1[j] = buf_load_f32<bf16>(rsv, (ok && lane + 64 < dk) ? (unsigned)((row + lane + 64) * 2) : EMO_OOB);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float pj = j0 + j < n ? prob[j0 + j] : 0.f;
        o0 += pj * v0[j];
        o1 += pj * v1[j];
      }
    }
  }
  const float inv = 1.f / l_run;
  if (lane < dk) out[lane] = (bf16)(o0 * inv);
  if (lane + 64 < dk) out[lane + 64] = (bf16)(o1 * inv);
}

// LDS layout shared by both kernels (bf16 rows of 16): x / h / o / y with d columns, qkv with 3d, act with F; per-wave
// scratch: q (128 floats) + prob (64 floats)
struct WgLds {
  bf16 *x, *h, *o, *y, *qkv, *act;
  float* wscr;
  int ldd, ldq, ldf;
};
__device__ WgLds wg_carve(char* smem, int d, int F) {
  WgLds L;
  L.ldd = d + 8; L.ldq = 3 * d + 8; L.ldf = F + 8;
  bf16* p = reinterpret_cast<bf16*>(smem);
  L.x = p; p += WG_ROWS * L.ldd;
  L.h = p; p += WG_ROWS * L.ldd;
  L.o = p; p += WG_ROWS * L.ldd;
  L.y = p; p += WG_ROWS * L.ldd;
  L.qkv = p; p += WG_ROWS * L.ldq;
  L.act = p; p += WG_ROWS * L.ldf;
  L.wscr = reinterpret_cast<float*>(p);
  return L;
}
size_t wg_lds_bytes(int d, int F) {
  return (size_t)2 * WG_ROWS * (4 * (d + 8) + (3 * d + 8) + (F + 8)) + (size_t)16 * 192 * 4;
}

__device__ __forceinline__ float gelu_(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

// Self-attention of the step for all (b, h) pairs: append the new k / v (from L.qkv) to the caches, then attend.
__device__ void wg_self_attention(const WgLds& L, int nb, int d, int H, int Lmax, bf16* kc, bf16* vc, int pos, int tid, int wave,
                                  int lane) {
  const int dk = d / H;
  for (int i = tid; i < nb * d; i += WG_THREADS) {
    const int b = i / d, c = i - b * d;
    kc[((long)b * Lmax + pos) * d + c] = L.qkv[b * L.ldq + d + c];
    vc[((long)b * Lmax + pos) * d + c] = L.qkv[b * L.ldq + 2 * d + c];
  }
  __threadfence_block();
  __syncthreads();   // the appended rows are read back from global memory by other waves of this workgroup
  float* q = L.wscr + wave * 192;
  float* prob = q + 128;
  for (int pair = wave; pair < nb * H; pair += 16) {
    const int b = pair / H, hh = pair - b * H;
    for (int c = lane; c < dk; c += 64) q[c] = (float)L.qkv[b * L.ldq + hh * dk + c];
    wg_attend(q, dk, kc + (long)b * Lmax * d + hh * dk, d, vc + (long)b * Lmax * d + hh * dk, d, pos + 1,
              1.f / sqrtf((float)dk), prob, L.o + b * L.ldd + hh * dk, lane);
  }
}

__global__ __launch_bounds__(WG_THREADS) void lm_step_wg_kernel(const LmWgArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const WgLds L = wg_carve(smem, a.d, a.F);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = a.d, nb = a.nb, pos = *a.pos;
  // rows >= nb of every activation buffer stay zero (their products are computed and never used)
  for (int i = tid; i < WG_ROWS * (4 * L.ldd + L.ldq + L.ldf); i += WG_THREADS) L.x[i] = (bf16)0.f;
  __syncthreads();
  // embeddings (word + position/token-type) -> y; x = LayerNorm(y)
  for (int i = tid; i < nb * d; i += WG_THREADS) {
    const int m = i / d, c = i - m * d;
    L.y[m * L.ldd + c] = (bf16)((float)a.word_emb[(long)a.ids[m] * d + c] + a.pe[(long)pos * d + c]);
  }
  __syncthreads();
  wg_layernorm(L.y, L.ldd, L.x, L.ldd, d, a.ln_emb, wave, lane);
  __syncthreads();
  const long layer_elems = (long)nb * a.Lmax * d;
  for (int li = 0; li < a.nl; ++li) {
    const emoasr_bert_layer_t& Ly = a.layers[li];
    wg_linear(L.x, L.ldd, d, (const bf16*)Ly.qkv.w, Ly.qkv.b, 3 * d, wave, lane, [&](int m, int n, float v) {
      L.qkv[m * L.ldq + n] = (bf16)v;
    });
    __syncthreads();
    wg_self_attention(L, nb, d, a.H, a.Lmax, a.kcache + li * layer_elems, a.vcache + li * layer_elems, pos, tid, wave, lane);
    __syncthreads();
    wg_linear(L.o, L.ldd, d, (const bf16*)Ly.attn_out.w, Ly.attn_out.b, d, wave, lane, [&](int m, int n, float v) {
      L.y[m * L.ldd + n] = (bf16)(v + (float)L.x[m * L.ldd + n]);
    });
    __syncthreads();
    wg_layernorm(L.y, L.ldd, L.x, L.ldd, d, Ly.ln_attn, wave, lane);
    __syncthreads();
    wg_linear(L.x, L.ldd, d, (const bf16*)Ly.inter.w, Ly.inter.b, a.F, wave, lane, [&](int m, int n, float v) {
      L.act[m * L.ldf + n] = (bf16)gelu_(v);
    });
    __syncthreads();
    wg_linear(L.act, L.ldf, a.F, (const bf16*)Ly.out.w, Ly.out.b, d, wave, lane, [&](int m, int n, float v) {
      L.y[m * L.ldd + n] = (bf16)(v + (float)L.x[m * L.ldd + n]);
    });
    __syncthreads();
    wg_layernorm(L.y, L.ldd, L.x, L.ldd, d, Ly.ln_out, wave, lane);
    __syncthreads();
  }
  wg_linear(L.x, L.ldd, d, (const bf16*)a.transform.w, a.transform.b, d, wave, lane, [&](int m, int n, float v) {
    if (m < nb) a.out_hidden[(long)m * d + n] = (bf16)gelu_(v);
  });
}

__global__ __launch_bounds__(WG_THREADS) void dec_step_wg_kernel(const DecWgArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const WgLds L = wg_carve(smem, a.d, a.F);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = a.d, nb = a.nb, pos = *a.pos, dk = a.d / a.H;
  for (int i = tid; i < WG_ROWS * (4 * L.ldd + L.ldq + L.ldf); i += WG_THREADS) L.x[i] = (bf16)0.f;
  __syncthreads();
  for (int i = tid; i < nb * d; i += WG_THREADS) {
    const int m = i / d, c = i - m * d;
    L.x[m * L.ldd + c] = (bf16)((float)a.embed[(long)a.ids[m] * d + c] * a.emb_scale + a.pe[(long)pos * d + c]);
  }
  __syncthreads();
  const long layer_elems = (long)nb * a.Lmax * d;
  for (int li = 0; li < a.nl; ++li) {
    const emoasr_decoder_layer_t& Ly = a.layers[li];
    // masked self-attention over the cached prefix
    wg_layernorm(L.x, L.ldd, L.h, L.ldd, d, Ly.ln1, wave, lane);
    __syncthreads();
    wg_linear(L.h, L.ldd, d, (const bf16*)Ly.qkv.w, Ly.qkv.b, 3 * d, wave, lane, [&](int m, int n, float v) {
      L.qkv[m * L.ldq + n] = (bf16)v;
    });
    __syncthreads();
    wg_self_attention(L, nb, d, a.H, a.Lmax, a.kcache + li * layer_elems, a.vcache + li * layer_elems, pos, tid, wave, lane);
    __syncthreads();
    wg_linear(L.o, L.ldd, d, (const bf16*)Ly.out.w, Ly.out.b, d, wave, lane, [&](int m, int n, float v) {
      L.x[m * L.ldd + n] = (bf16)(v + (float)L.x[m * L.ldd + n]);
    });
    __syncthreads();
    // source attention against the encoder memory's cached K | V
    wg_layernorm(L.x, L.ldd, L.h, L.ldd, d, Ly.ln2, wave, lane);
    __syncthreads();
    wg_linear(L.h, L.ldd, d, (const bf16*)Ly.q2.w, Ly.q2.b, d, wave, lane, [&](int m, int n, float v) {
      L.qkv[m * L.ldq + n] = (bf16)v;
    });
    __syncthreads();
    {
      float* q = L.wscr + wave * 192;
      float* prob = q + 128;
      const bf16* kvl = static_cast<const bf16*>(a.kv[li]);
      for (int pair = wave; pair < nb * a.H; pair += 16) {
        const int b = pair / a.H, hh = pair - b * a.H;
        for (int c = lane; c < dk; c += 64) q[c] = (float)L.qkv[b * L.ldq + hh * dk + c];
        const bf16* kb = kvl + (long)b * a.T * 2 * d + hh * dk;
        wg_attend(q, dk, kb, 2 * d, kb + d, 2 * d, min(a.kmem[b], a.T), 1.f / sqrtf((float)dk), prob,
                  L.o + b * L.ldd + hh * dk, lane);
      }
    }
    __syncthreads();
    wg_linear(L.o, L.ldd, d, (const bf16*)Ly.out2.w, Ly.out2.b, d, wave, lane, [&](int m, int n, float v) {
      L.x[m * L.ldd + n] = (bf16)(v + (float)L.x[m * L.ldd + n]);
    });
    __syncthreads();
    // feed-forward
    wg_layernorm(L.x, L.ldd, L.h, L.ldd, d, Ly.ln3, wave, lane);
    __syncthreads();
    wg_linear(L.h, L.ldd, d, (const bf16*)Ly.w1.w, Ly.w1.b, a.F, wave, lane, [&](int m, int n, float v) {
      L.act[m * L.ldf + n] = (bf16)fmaxf(v, 0.f);
    });
    __syncthreads();
    wg_linear(L.act, L.ldf, a.F, (const bf16*)Ly.w2.w, Ly.w2.b, d, wave, lane, [&](int m, int n, float v) {
      L.x[m * L.ldd + n] = (bf16)(v + (float)L.x[m * L.ldd + n]);
    });
    __syncthreads();
  }
  for (int i = tid; i < nb * d; i += WG_THREADS) {
    const int m = i / d, c = i - m * d;
    a.out_x[(long)m * d + c] = L.x[m * L.ldd + c];
  }
}

int g_decode_wg = 0;  // measured slower than the launch chain (see the header): off unless emoasr_set_option("decode_wg", 1)

template <typename K>
int set_lds(K kernel, size_t bytes) {
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) { emo_set_error("hipFuncSetAttribute(%zu): %s", bytes, hipGetErrorString(e)); return 1; }
  return 0;
}

}  // namespace

void emo_decode_set_wg(int v) { g_decode_wg = v; }

// Can the single-workgroup kernels take this step?  (bf16, <= 16 hypotheses, <= 12 / 8 layers, widths that fit the LDS plan)
bool emo_decode_wg_ok(int dtype, int nb, int nl, int max_layers, int d, int H, int F) {
  return g_decode_wg && dtype == EMO_BF16 && nb >= 1 && nb <= WG_ROWS && nl >= 1 && nl <= max_layers && d % 32 == 0 && F % 32 == 0 &&
         d <= 512 && F <= WG_MAXD && d % H == 0 && (d / H) % 8 == 0 && d / H <= 128 && wg_lds_bytes(d, F) <= 160 * 1024;
}

// The LM stack up to GELU(transform(x)) -> out_hidden [nb, d] (bf16); the caller applies ln_transform + the tied projection.
int emo_bert_lm_step_wg(int nl, const emoasr_bert_layer_t* layers, const emoasr_bert_step_t* io, void* out_hidden, hipStream_t s) {
  LmWgArgs a{};
  a.nl = nl; a.nb = io->nb; a.Lmax = io->Lmax; a.d = io->d; a.H = io->H; a.F = io->F;
  a.ids = io->ids; a.pos = io->pos; a.word_emb = (const bf16*)io->word_emb; a.pe = io->pe; a.ln_emb = io->ln_emb;
  a.kcache = (bf16*)io->kcache; a.vcache = (bf16*)io->vcache; a.transform = io->transform; a.out_hidden = (bf16*)out_hidden;
  for (int i = 0; i < nl; ++i) a.layers[i] = layers[i];
  const size_t bytes = wg_lds_bytes(a.d, a.F);
  static size_t set_bytes = 0;
  if (bytes > set_bytes) { if (set_lds(lm_step_wg_kernel, bytes)) return 1; set_bytes = bytes; }
  lm_step_wg_kernel<<<1, WG_THREADS, bytes, s>>>(a);
  EMO_LAUNCH_CHECK();
  return 0;
}

// The decoder stack up to (not including) the final LayerNorm -> out_x [nb, dd] (bf16).
int emo_transformer_decoder_step_wg(int nl, const emoasr_decoder_layer_t* layers, const emoasr_decoder_step_t* io, void* out_x,
                                    hipStream_t s) {
  DecWgArgs a{};
  a.nl = nl; a.nb = io->nb; a.Lmax = io->Lmax; a.T = io->T; a.d = io->dd; a.H = io->H; a.F = io->F;
  a.ids = io->ids; a.pos = io->pos; a.embed = (const bf16*)io->embed; a.pe = io->pe; a.emb_scale = io->emb_scale;
  a.kcache = (bf16*)io->kcache; a.vcache = (bf16*)io->vcache; a.kmem = io->kmem; a.out_x = (bf16*)out_x;
  for (int i = 0; i < nl; ++i) { a.layers[i] = layers[i]; a.kv[i] = io->kv[i]; }
  const size_t bytes = wg_lds_bytes(a.d, a.F);
  static size_t set_bytes = 0;
  if (bytes > set_bytes) { if (set_lds(dec_step_wg_kernel, bytes)) return 1; set_bytes = bytes; }
  dec_step_wg_kernel<<<1, WG_THREADS, bytes, s>>>(a);
  EMO_LAUNCH_CHECK();
  return 0;
}
