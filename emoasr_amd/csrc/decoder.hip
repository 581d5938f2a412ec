// Transformer-decoder side kernels: token embedding (+ sqrt(d) scale, absolute positional
// encoding, dropout) forward/backward, and the label-smoothing cross-entropy (loss + gradient).
//
// Reference: asr/modeling/decoders/transformer.py:99 (self.pe(self.embed(ys_in))),
// asr/modeling/transformer.py:43-45, asr/criteria.py:5-46 (LabelSmoothingLoss: 1-eps on the label,
// eps/(V-1) on every other class, summed over t < ylens[b], optional /ylen and /B).
#include "common.h"
#include "../../include/emoasr_hip.h"

namespace {

// out[m, :] = (table[ids[m], :] * scale + pe[m % L, :]) * dropout(seed, m*d + c)
template <typename T>
__global__ __launch_bounds__(256) void embed_fwd_kernel(int M, int L, int d, const int* __restrict__ ids,
                                                        const T* __restrict__ table,
                                                        const float* __restrict__ pe, float scale, float p,
                                                        uint64_t seed, T* __restrict__ out) {
  const long n = (long)M * d;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = i % d;
    const long m = i / d;
    float v = to_f32(table[(long)ids[m] * d + c]) * scale;
    if (pe) v += pe[(m % L) * d + c];
    out[i] = from_f32<T>(v * dropout_scale(seed, (uint64_t)i, p));
  }
}

// dtable[ids[m], :] += dout[m, :] * scale * dropout(seed, m*d + c)
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_kernel(int M, int d, const int* __restrict__ ids,
                                                        const T* __restrict__ dout, float scale, float p,
                                                        uint64_t seed, float* __restrict__ dtable) {
  const long n = (long)M * d;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = i % d;
    const long m = i / d;
    atomicAdd(&dtable[(long)ids[m] * d + c], to_f32(dout[i]) * scale * dropout_scale(seed, (uint64_t)i, p));
  }
}

// One block per row m.  w[m] = 0 for padded positions, else the row weight (1/B, /ylen).
//   loss[m] = -w * ((1-eps) * logp[y] + eps/(V-1) * (sum_v logp[v] - logp[y]))
//   grad[m, v] = gscale * w * (softmax[v] - q[v])        (sum_v q = 1)
template <typename T>
__global__ __launch_bounds__(256) void lsm_loss_kernel(int V, const T* __restrict__ logits, long ld,
                                                       const int* __restrict__ labels,
                                                       const float* __restrict__ w, float eps,
                                                       float* __restrict__ loss, float gscale,
                                                       const float* __restrict__ gscale_dev,
                                                       T* __restrict__ grad, long ldg) {
  __shared__ float red[16];
  const long m = blockIdx.x;
  const T* row = logits + m * ld;
  const float wm = w[m];
  if (wm == 0.f) {
    if (threadIdx.x == 0) loss[m] = 0.f;
    if (grad) for (int v = threadIdx.x; v < V; v += 256) grad[m * ldg + v] = from_f32<T>(0.f);
    return;
  }
  float mx = -INFINITY, sum = 0.f;
  for (int v = threadIdx.x; v < V; v += 256) { const float x = to_f32(row[v]); mx = fmaxf(mx, x); sum += x; }
  mx = block_max(mx, red);
  sum = block_sum(sum, red);
  float se = 0.f;
  for (int v = threadIdx.x; v < V; v += 256) se += __expf(to_f32(row[v]) - mx);
  se = block_sum(se, red);
  const float lse = mx + logf(se);
  const int y = labels[m];
  const float lpy = to_f32(row[y]) - lse;
  const float off = eps / (V - 1);
  if (threadIdx.x == 0) loss[m] = -wm * ((1.f - eps) * lpy + off * (sum - V * lse - lpy));
  if (grad) {
    const float gs = (gscale_dev ? gscale * gscale_dev[0] : gscale) * wm;
    for (int v = threadIdx.x; v < V; v += 256) {
      const float pv = __expf(to_f32(row[v]) - lse);
      grad[m * ldg + v] = from_f32<T>(gs * (pv - (v == y ? 1.f - eps : off)));
    }
  }
}

inline int ew_grid(long n) { long b = (n + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

}  // namespace

extern "C" int emoasr_embed_fwd(int dtype, int M, int L, int d, const int* ids, const void* table,
                                const float* pe, float scale, float drop_p, uint64_t seed, void* out,
                                void* stream) {
  if (M == 0) return 0;
  EMO_DISPATCH(dtype, (embed_fwd_kernel<T><<<ew_grid((long)M * d), 256, 0, (hipStream_t)stream>>>(
                          M, L, d, ids, (const T*)table, pe, scale, drop_p, seed, (T*)out)));
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_embed_bwd(int dtype, int M, int d, const int* ids, const void* dout, float scale,
                                float drop_p, uint64_t seed, float* dtable, void* stream) {
  if (M == 0) return 0;
  EMO_DISPATCH(dtype, (embed_bwd_kernel<T><<<ew_grid((long)M * d), 256, 0, (hipStream_t)stream>>>(
                          M, d, ids, (const T*)dout, scale, drop_p, seed, dtable)));
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_lsm_loss(int dtype, int M, int V, const void* logits, long ld, const int* labels,
                               const float* w, float lsm_prob, float* loss, float gscale,
                               const float* gscale_dev, void* grad, long ldg, void* stream) {
  if (M == 0) return 0;
  EMO_CHECK(V >= 2, "lsm_loss: V=%d", V);
  EMO_DISPATCH(dtype, (lsm_loss_kernel<T><<<M, 256, 0, (hipStream_t)stream>>>(
                          V, (const T*)logits, ld, labels, w, lsm_prob, loss, gscale, gscale_dev, (T*)grad, ldg)));
  EMO_LAUNCH_CHECK();
  return 0;
}

// =======================================================================================
// Beam-search side kernels (asr/modeling/decoders/transformer.py:161-294, ctc_score.py:13-85)
// =======================================================================================
namespace {

// out[m, v] = log_softmax(x[m, :V])[v] + mu * add[m, v]        (add optional, row stride lda)
template <typename T>
__global__ __launch_bounds__(256) void log_softmax_kernel(int V, const T* __restrict__ x, long ldx,
                                                          const float* __restrict__ add, long lda, float mu,
                                                          float* __restrict__ out, long ldo) {
  __shared__ float red[16];
  const long m = blockIdx.x;
  const T* row = x + m * ldx;
  float mx = -INFINITY;
  for (int v = threadIdx.x; v < V; v += 256) mx = fmaxf(mx, to_f32(row[v]));
  mx = block_max(mx, red);
  float se = 0.f;
  for (int v = threadIdx.x; v < V; v += 256) se += expf(to_f32(row[v]) - mx);
  se = block_sum(se, red);
  const float lse = mx + logf(se);
  for (int v = threadIdx.x; v < V; v += 256) {
    float r = to_f32(row[v]) - lse;
    if (add) r += mu * add[m * lda + v];
    out[m * ldo + v] = r;
  }
}

// k largest of each row, descending, ties -> lowest index.  k <= 64.  One block per row: k rounds of
// block-wide arg-max over a private copy in LDS (V floats).
__global__ __launch_bounds__(256) void topk_kernel(int V, int k, const float* __restrict__ x, long ldx,
                                                   const float* __restrict__ aux, long ldaux,
                                                   float* __restrict__ vals, int* __restrict__ idx,
                                                   float* __restrict__ aux_out) {
  extern __shared__ float buf[];  // [V]
  __shared__ float rv[4];
  __shared__ int ri[4];
  const long m = blockIdx.x;
  for (int v = threadIdx.x; v < V; v += 256) buf[v] = x[m * ldx + v];
  __syncthreads();
  for (int j = 0; j < k; ++j) {
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int v = threadIdx.x; v < V; v += 256) {
      const float c = buf[v];
      if (c > best || (c == best && v < bi)) { best = c; bi = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { rv[threadIdx.x >> 6] = best; ri[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < 4; ++w)
        if (rv[w] > best || (rv[w] == best && ri[w] < bi)) { best = rv[w]; bi = ri[w]; }
      if (bi == 0x7fffffff) bi = 0;
      vals[m * k + j] = best;
      idx[m * k + j] = bi;
      if (aux_out) aux_out[m * k + j] = aux[m * ldaux + bi];
      buf[bi] = -INFINITY;
    }
    __syncthreads();
  }
}

#define EMO_DPP_I(old_, v_, ctrl_, rmask_) __builtin_amdgcn_update_dpp((old_), (v_), (ctrl_), (rmask_), 0xf, false)
// wave-wide max / min through DPP moves: quad swaps, half-row and row mirrors, the GFX9 row broadcasts; the result of lane 63
__device__ __forceinline__ float topk_wave_max(float v) {
#define EMO_STEP(ctrl_, rm_) v = fmaxf(v, __builtin_bit_cast(float, EMO_DPP_I(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), ctrl_, rm_)))
  EMO_STEP(0xB1, 0xf); EMO_STEP(0x4E, 0xf); EMO_STEP(0x141, 0xf); EMO_STEP(0x140, 0xf); EMO_STEP(0x142, 0xa); EMO_STEP(0x143, 0xc);
#undef EMO_STEP
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ int topk_wave_min(int v) {
#define EMO_STEP(ctrl_, rm_) v = min(v, EMO_DPP_I(v, v, ctrl_, rm_))
  EMO_STEP(0xB1, 0xf); EMO_STEP(0x4E, 0xf); EMO_STEP(0x141, 0xf); EMO_STEP(0x140, 0xf); EMO_STEP(0x142, 0xa); EMO_STEP(0x143, 0xc);
#undef EMO_STEP
  return __builtin_amdgcn_readlane(v, 63);
}

// log-softmax of the decoder row + mu * log-softmax of the LM row + top-k, one launch, one workgroup of 1024 threads per
// hypothesis (the beam search never needs the full score rows: three launches and ~130 us per step become one of ~20 us).
//   s[v] = (dec[v] - lse(dec)) + mu * (lm[v] - lse(lm));  vals / idx = the k largest s (ties -> lowest index);
//   lm_at = lm[idx] - lse(lm).   lm == NULL: no LM term.
// Selection: every thread keeps the best of its own strided elements; a round is one wave reduction + one barrier, then
// only the owner of the winner rescans its elements.
template <typename T>
__global__ __launch_bounds__(1024) void beam_scores_topk_kernel(int V, int k, const T* __restrict__ dec, long ldd,
                                                                const float* __restrict__ lm, long ldl, float mu,
                                                                float* __restrict__ vals, int* __restrict__ idx,
                                                                float* __restrict__ lm_at, int lm_lds) {
  extern __shared__ float sbuf[];      // [V] the decoder row (f32), later the scores | [16][k] x 2 survivors | (lm_lds) [V] the LM row
  __shared__ float red4[4][16];
  const long m = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* drow = dec + m * ldd;
  const float* lrow = lm ? lm + m * ldl : nullptr;
  // Both rows go to LDS first, four strided elements of each per thread and trip (8 loads in flight): read where they are used,
  // inside three loops of ~V / 1024 dependent trips each, the rows cost ~30 global round trips -- most of the kernel's 41 us.
  // A thread only ever reads back the elements it staged itself (v = tid mod 1024), so no barrier is needed.
  float* slm = (lrow && lm_lds) ? sbuf + V + 32 * k : nullptr;
  for (int v0 = tid; v0 < V; v0 += 4096) {
    float a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int v = v0 + 1024 * u;
      a[u] = v < V ? to_f32(drow[v]) : 0.f;
      b[u] = (slm && v < V) ? lrow[v] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int v = v0 + 1024 * u;
      if (v < V) { sbuf[v] = a[u]; if (slm) slm[v] = b[u]; }
    }
  }
  const float* lsrc = slm ? slm : lrow;   // (LM row too long for LDS: read from global memory as before)
  // ---- the two log-sum-exps ----
  float mxd = -INFINITY, mxl = -INFINITY;
  for (int v = tid; v < V; v += 1024) {
    mxd = fmaxf(mxd, sbuf[v]);
    if (lrow) mxl = fmaxf(mxl, lsrc[v]);
  }
  for (int o = 32; o > 0; o >>= 1) { mxd = fmaxf(mxd, __shfl_xor(mxd, o)); mxl = fmaxf(mxl, __shfl_xor(mxl, o)); }
  if (lane == 0) { red4[0][wave] = mxd; red4[1][wave] = mxl; }
  __syncthreads();
  mxd = red4[0][0]; mxl = red4[1][0];
  for (int w = 1; w < 16; ++w) { mxd = fmaxf(mxd, red4[0][w]); mxl = fmaxf(mxl, red4[1][w]); }
  float sd = 0.f, sl = 0.f;
  for (int v = tid; v < V; v += 1024) {
    sd += expf(sbuf[v] - mxd);
    if (lrow) sl += expf(lsrc[v] - mxl);
  }
  for (int o = 32; o > 0; o >>= 1) { sd += __shfl_xor(sd, o); sl += __shfl_xor(sl, o); }
  if (lane == 0) { red4[2][wave] = sd; red4[3][wave] = sl; }
  __syncthreads();
  sd = 0.f; sl = 0.f;
  for (int w = 0; w < 16; ++w) { sd += red4[2][w]; sl += red4[3][w]; }
  const float lsed = mxd + logf(sd), lsel = lrow ? mxl + logf(sl) : 0.f;
  // ---- scores; local best of this thread's elements ----
  // Two levels, so that a round costs no workgroup barrier: every wave selects the k best of its own lanes' elements (a round = two
  // DPP reductions -- the maximum, then the lowest index among the lanes that hold it -- and a rescan of the lanes' elements),
  // then wave 0 merges the 16 x k survivors the same way.  The elements being rescanned sit in REGISTERS (V <= 16 K, k <= 32;
  // beyond that the LDS copies are walked): a rescan loop over LDS with a run-time trip count waits ~130 cycles per element for
  // one useful lane, which -- not the reductions, the barrier or the global loads -- was 2 us per round (42 us at k = 15).
  constexpr int NE = 16, NM = 8;
  const bool e_regs = V <= NE * 1024;
  float lb = -INFINITY; int li = 0x7fffffff;
  float e[NE];
  if (e_regs) {
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      const int v = tid + 1024 * u;
      e[u] = -INFINITY;
      if (v < V) {
        float r = sbuf[v] - lsed;
        if (lrow) r += mu * (lsrc[v] - lsel);
        e[u] = r;
        if (r > lb) { lb = r; li = v; }   // ascending v: ties keep the lower index
      }
    }
  } else {
    for (int v = tid; v < V; v += 1024) {
      float r = sbuf[v] - lsed;
      if (lrow) r += mu * (lsrc[v] - lsel);
      sbuf[v] = r;
      if (r > lb) { lb = r; li = v; }
    }
  }
  float* cand_v = sbuf + V;                              // [16][k] survivors of the waves (dynamic LDS after the score row)
  int* cand_i = reinterpret_cast<int*>(cand_v + 16 * k);
  for (int j = 0; j < k; ++j) {
    const float best = topk_wave_max(lb);
    const int bi = topk_wave_min(lb == best ? li : 0x7fffffff);
    if (lane == 0) { cand_v[wave * k + j] = best; cand_i[wave * k + j] = bi; }
    if (e_regs) {   // every lane: drop the winner if it is mine, take the best of what is left
      lb = -INFINITY; li = 0x7fffffff;
#pragma unroll
      for (int u = 0; u < NE; ++u) {
        if (tid + 1024 * u == bi) e[u] = -INFINITY;
        if (e[u] > lb) { lb = e[u]; li = tid + 1024 * u; }
      }
    } else if (bi != 0x7fffffff && (bi & 1023) == tid) {
      sbuf[bi] = -INFINITY;
      lb = -INFINITY; li = 0x7fffffff;
      for (int v = tid; v < V; v += 1024) {
        const float c = sbuf[v];
        if (c > lb) { lb = c; li = v; }
      }
    }
  }
  __syncthreads();
  if (wave == 0) {
    // lane owns survivors lane, lane + 64, ... of the 16 k (indices are unique; -inf / 0x7fffffff entries are never picked)
    const int n = 16 * k;
    const bool m_regs = n <= NM * 64;
    float cv[NM]; int ci[NM];
#pragma unroll
    for (int u = 0; u < NM; ++u) {
      const int c = lane + 64 * u;
      cv[u] = (m_regs && c < n) ? cand_v[c] : -INFINITY;
      ci[u] = (m_regs && c < n) ? cand_i[c] : 0x7fffffff;
    }
    float mb = -INFINITY; int mi = 0x7fffffff;
    auto rescan = [&]() {
      mb = -INFINITY; mi = 0x7fffffff;
      if (m_regs) {
#pragma unroll
        for (int u = 0; u < NM; ++u)
          if (cv[u] > mb || (cv[u] == mb && ci[u] < mi)) { mb = cv[u]; mi = ci[u]; }
      } else {
        for (int c = lane; c < n; c += 64) {
          const float xv = cand_v[c];
          const int xi = cand_i[c];
          if (xv > mb || (xv == mb && xi < mi)) { mb = xv; mi = xi; }
        }
      }
    };
    rescan();
    for (int j = 0; j < k; ++j) {
      const float best = topk_wave_max(mb);
      int bi = topk_wave_min(mb == best ? mi : 0x7fffffff);
      if (bi != 0x7fffffff) {
        if (m_regs) {
#pragma unroll
          for (int u = 0; u < NM; ++u)
            if (ci[u] == bi) { cv[u] = -INFINITY; ci[u] = 0x7fffffff; }
        } else {
          for (int c = lane; c < n; c += 64)
            if (cand_i[c] == bi) { cand_v[c] = -INFINITY; cand_i[c] = 0x7fffffff; }
        }
        rescan();
      }
      if (bi == 0x7fffffff) bi = 0;
      if (lane == 0) {
        vals[m * k + j] = best;
        idx[m * k + j] = bi;
        if (lm_at) lm_at[m * k + j] = lrow ? lsrc[bi] - lsel : 0.f;   // (from the LDS copy: a global load here was a round trip per round)
      }
    }
  }
}

#define EMO_LOG0 (-1e10f)
__device__ __forceinline__ float np_logaddexp(float a, float b) {
  // numpy.logaddexp semantics for finite inputs
  const float d = a - b;
  return d > 0.f ? a + log1pf(expf(-d)) : b + log1pf(expf(d));
}

// One wave per (beam m, candidate c).  x: CTC log-probs [T, V].
// r_prev: the parent's state [T,2]: prev_states + (parent[m]*cw_prev + pcand[m]) * T*2, or init_state
// when prev_states == NULL.  Outputs log_psi [nb, cw] and states [nb, cw, T, 2].
// The recursion over t is sequential (and kept in the reference's float32 operation order: ctc_score.py:47-76),
// but everything it consumes is not: per chunk of 64 frames the lanes fetch the parent state and the two
// emission columns (one memory round trip per chunk instead of four per frame) and compute phi; lane 0 then
// runs the 64 dependent steps out of LDS and the lanes write the new state rows back coalesced.
__global__ __launch_bounds__(64) void ctc_prefix_kernel(int Tn, int V, int cw, const float* __restrict__ x,
                                                        const float* __restrict__ prev_states, int cw_prev,
                                                        const int* __restrict__ parent, const int* __restrict__ pcand,
                                                        const float* __restrict__ init_state,
                                                        const int* __restrict__ last, const int* __restrict__ out_len,
                                                        const int* __restrict__ cands, int blank, int eos,
                                                        float* __restrict__ log_psi, float* __restrict__ states) {
  __shared__ float s_rn[64], s_rb[64];
  const int m = blockIdx.x / cw, c = blockIdx.x % cw, lane = threadIdx.x;
  const float* rp = prev_states ? prev_states + ((long)parent[m] * cw_prev + pcand[m]) * Tn * 2 : init_state;
  const int tok = cands[m * cw + c];
  const int olen = out_len[m];
  const bool same = olen > 0 && tok == last[m];
  float* r = states + ((long)m * cw + c) * Tn * 2;
  const int start = olen > 1 ? olen : 1;
  // rows before `start` keep their initial value: LOG_0, except r[0][0] = x[0][tok] for an empty prefix
  for (int t = lane; t < start && t < Tn; t += 64) {
    r[2 * t] = (t == 0 && olen == 0) ? x[tok] : EMO_LOG0;
    r[2 * t + 1] = EMO_LOG0;
  }
  // The three recursions have one shape, v <- logaddexp(v, a_t) + b_t:
  //   lane 0: r^n   a = phi_t            b = x_t[tok]
  //   lane 1: psi   a = phi_t + x_t[tok] b = 0
  //   lane 2: r^b   a = r^n_{t-1}        b = x_t[blank]     (a comes from lane 0, one step behind)
  // so lanes 0..2 run them in lockstep: one logaddexp per frame on the critical path instead of three.
  const float v0 = (start == 1 && olen == 0) ? x[tok] : EMO_LOG0;
  float v = lane == 2 ? EMO_LOG0 : v0;  // rn = psi = v0, rb = LOG_0
  for (int t0 = start; t0 < Tn; t0 += 64) {
    const int t = t0 + lane;
    // lane i keeps frame t0 + i's inputs in registers; the recurrence lanes fetch them with v_readlane (the LDS copies they used
    // to read cost a ~130-cycle round trip per frame on top of the dependent exp / log chain: 53 us at T' 285)
    float my_phi = 0.f, my_x = 0.f, my_xb = 0.f;
    if (t < Tn) {
      const float pn = rp[2 * (t - 1)], pb = rp[2 * (t - 1) + 1];
      my_phi = same ? pb : np_logaddexp(pn, pb);
      my_x = x[(long)t * V + tok];
      my_xb = x[(long)t * V + blank];
    }
    const int n = min(64, Tn - t0);
    for (int i = 0; i < n; ++i) {
      const float rn_prev = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
      const float phi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_phi), i));
      const float xt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_x), i));
      const float xbl = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_xb), i));
      const float a = lane == 0 ? phi : (lane == 1 ? phi + xt : rn_prev);
      const float bb = lane == 0 ? xt : (lane == 1 ? 0.f : xbl);
      if (lane < 3) {
        // same formula as np_logaddexp with the hardware exp / log (v_exp_f32, v_log_f32): |error| < 1e-7 in
        // absolute terms per step against scores of magnitude 10..100 (the library expf / log1pf pair is ~4x
        // the latency of this loop's critical path)
        const float d = v - a;
        v = (d > 0.f ? v + __logf(1.f + __expf(-d)) : a + __logf(1.f + __expf(d))) + bb;
      }
      if (lane == 0) s_rn[i] = v;
      if (lane == 2) s_rb[i] = v;
    }
    __builtin_amdgcn_wave_barrier();
    if (t < Tn) { r[2 * t] = s_rn[lane]; r[2 * t + 1] = s_rb[lane]; }
    __builtin_amdgcn_wave_barrier();
  }
  float psi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 1));
  if (lane == 0) {
    if (tok == eos) psi = np_logaddexp(rp[2 * (Tn - 1)], rp[2 * (Tn - 1) + 1]);
    if (tok == blank) psi = EMO_LOG0;
    log_psi[m * cw + c] = psi;
  }
}

// initial state: r[t,1] = running sum of the blank log-probs (float32, step by step), r[t,0] = LOG_0
__global__ void ctc_prefix_init_kernel(int Tn, int V, const float* __restrict__ x, int blank,
                                       float* __restrict__ r) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float acc = 0.f;
  for (int t = 0; t < Tn; ++t) {
    acc = t == 0 ? x[blank] : acc + x[(long)t * V + blank];
    r[2 * t] = EMO_LOG0;
    r[2 * t + 1] = acc;
  }
}

}  // namespace

extern "C" int emoasr_log_softmax(int dtype, int M, int V, const void* x, long ldx, const float* add, long lda,
                                  float mu, float* out, long ldo, void* stream) {
  if (M == 0) return 0;
  EMO_DISPATCH(dtype, (log_softmax_kernel<T><<<M, 256, 0, (hipStream_t)stream>>>(V, (const T*)x, ldx, add, lda, mu,
                                                                                out, ldo)));
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_topk(int M, int V, int k, const float* x, long ldx, const float* aux, long ldaux,
                           float* vals, int* idx, float* aux_out, void* stream) {
  if (M == 0) return 0;
  EMO_CHECK(k >= 1 && k <= V, "topk: k=%d V=%d", k, V);
  EMO_CHECK((size_t)V * 4 <= 150 * 1024, "topk: V=%d too large for an LDS row", V);
  if ((size_t)V * 4 > 60 * 1024)
    hipFuncSetAttribute((const void*)topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, V * 4);
  topk_kernel<<<M, 256, sizeof(float) * V, (hipStream_t)stream>>>(V, k, x, ldx, aux, ldaux, vals, idx, aux_out);
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_beam_scores_topk(int dtype, int M, int V, int k, const void* dec, long ldd, const float* lm, long ldl,
                                       float mu, float* vals, int* idx, float* lm_at, void* stream) {
  if (M == 0) return 0;
  EMO_CHECK(k >= 1 && k <= V, "beam_scores_topk: k=%d V=%d", k, V);
  int bytes = V * 4 + 16 * k * 8;   // the score row + the 16 waves' k survivors (value, index)
  EMO_CHECK((size_t)bytes <= 150 * 1024, "beam_scores_topk: V=%d, k=%d too large for the LDS plan", V, k);
  const int lm_lds = lm && (size_t)bytes + (size_t)V * 4 <= 150 * 1024;   // the LM row staged in LDS as well
  if (lm_lds) bytes += V * 4;
  if (dtype == EMO_BF16) {
    if (bytes > 60 * 1024) hipFuncSetAttribute((const void*)beam_scores_topk_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    beam_scores_topk_kernel<bf16><<<M, 1024, bytes, (hipStream_t)stream>>>(V, k, (const bf16*)dec, ldd, lm, ldl, mu, vals, idx, lm_at, lm_lds);
  } else if (emo_is_f32(dtype)) {
    if (bytes > 60 * 1024) hipFuncSetAttribute((const void*)beam_scores_topk_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    beam_scores_topk_kernel<float><<<M, 1024, bytes, (hipStream_t)stream>>>(V, k, (const float*)dec, ldd, lm, ldl, mu, vals, idx, lm_at, lm_lds);
  } else {
    emo_set_error("bad dtype %d", dtype);
    return 1;
  }
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_ctc_prefix_init(int T, int V, const float* x, int blank, float* r, void* stream) {
  ctc_prefix_init_kernel<<<1, 64, 0, (hipStream_t)stream>>>(T, V, x, blank, r);
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_ctc_prefix_score(int nb, int T, int V, int cw, const float* x, const float* prev_states,
                                       int cw_prev, const int* parent, const int* pcand, const float* init_state,
                                       const int* last, const int* out_len, const int* cands, int blank, int eos,
                                       float* log_psi, float* states, void* stream) {
  if (nb == 0) return 0;
  EMO_CHECK(cw >= 1 && cw <= 1024, "ctc_prefix_score: cw=%d", cw);
  EMO_CHECK(prev_states || init_state, "ctc_prefix_score: no previous state");
  ctc_prefix_kernel<<<nb * cw, 64, 0, (hipStream_t)stream>>>(T, V, cw, x, prev_states, cw_prev, parent, pcand,
                                                                     init_state, last, out_len, cands, blank, eos,
                                                                     log_psi, states);
  EMO_LAUNCH_CHECK();
  return 0;
}
