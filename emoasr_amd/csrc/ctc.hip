// CTC: row log-sum-exp over the vocabulary, forward/backward lattices, gradient
// w.r.t. the logits, and greedy decoding.
//
// Reference semantics: asr/modeling/decoders/ctc.py:36-38,109-113 ->
//   nn.CTCLoss(blank, reduction="sum", zero_infinity=True)(log_softmax(logits), ys, elens, ylens) / B
// and ctc.py:176-201 (_greedy: raw-logit argmax, first max wins, collapse repeats,
// drop blanks, <eos> kept).
//
// Lattice layout: extended label sequence l' = [blank, y1, blank, y2, ..., blank] with
// S' = 2*ylen+1 states per utterance, padded to S = 2*Lmax+1.  The emissions
// lp[b,t,s] = logits[b,t,l'_s] - lse[b,t] are gathered once by a fully parallel
// kernel; the two sequential scans (alpha forward, beta backward in time) then stream
// contiguous S-wide rows.  One block per (utterance, direction): states across lanes,
// neighbour exchange through LDS, one barrier per frame.
#include <algorithm>
#include "common.h"
#include "../../include/emoasr_hip.h"

namespace {

// log-space sums on the hardware exp2 / log2 units (v_exp_f32 / v_log_f32 through __expf / __logf): the
// lattice recursions are one dependent chain per frame, and the accurate expf / logf / log1pf expansions
// (~20 instructions each) were a large part of a step.  Absolute error ~1e-6 per step.  (A one-wave variant
// with two states per lane and shuffles instead of LDS + barrier was measured no faster: the step is bound
// by the dependent max / exp / log chain, ~0.33 us.)
__device__ __forceinline__ float lat_add(float a, float b) {
  const float m = fmaxf(a, b);
  if (m == -INFINITY) return -INFINITY;
  return m + __logf(1.f + __expf(-fabsf(a - b)));
}
__device__ __forceinline__ float lat_add3(float a, float b, float c) {
  const float m = fmaxf(a, fmaxf(b, c));
  if (m == -INFINITY) return -INFINITY;
  return m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
}

template <typename T>
__global__ __launch_bounds__(256) void row_lse_kernel(int V, const T* __restrict__ logits, long ld,
                                                      float* __restrict__ lse, int aligned) {
  __shared__ float red[16];
  const T* row = logits + (long)blockIdx.x * ld;
  constexpr int VEC = 16 / sizeof(T);
  float m = -INFINITY, s = 0.f;
  const int nv = aligned ? V / VEC : 0;
  for (int i = threadIdx.x; i < nv; i += 256) {
    const Vec16<T> v = load16(row + (long)i * VEC);
    float lm = v.get(0);
#pragma unroll
    for (int j = 1; j < VEC; ++j) lm = fmaxf(lm, v.get(j));
    if (lm > m) { s *= __expf(m - lm); m = lm; }
#pragma unroll
    for (int j = 0; j < VEC; ++j) s += __expf(v.get(j) - m);
  }
  for (int i = nv * VEC + threadIdx.x; i < V; i += 256) {
    const float x = to_f32(row[i]);
    if (x > m) { s *= __expf(m - x); m = x; }
    s += __expf(x - m);
  }
  const float gm = block_max(m, red);
  const float gs = block_sum(m == -INFINITY ? 0.f : s * __expf(m - gm), red);
  if (threadIdx.x == 0) lse[blockIdx.x] = gm + logf(gs);
}

__device__ __forceinline__ int ext_label(const int* lab, int s, int blank) {
  return (s & 1) ? lab[s >> 1] : blank;
}

// Stacked micro-batches (engine.ctc_train_stacked): the utterances of several micro-batches in ONE set of launches.  Their logits
// rows are not b * Tn + t then: utterance b starts at row row0[b] and has tpad[b] rows (its micro-batch's padded length); the
// gradient scale is per utterance (uscale[b] = weight of its micro-batch / its micro-batch's size).  The lattice tables stay
// [B, Tn, S] with Tn = the longest padded length.  All three NULL: the dense [B, Tn, V] layout.
struct UttRows { const long* row0; const int* tpad; const float* uscale; };

// lp[b,t,s] = logits[b,t,l'_s] - lse[b,t]   (t < elens[b], s < 2*ylens[b]+1; else -inf)
template <typename T>
__global__ __launch_bounds__(256) void ctc_gather_kernel(int Tn, int S, int Lmax, const T* __restrict__ logits,
                                                         long ld, const float* __restrict__ lse,
                                                         const int* __restrict__ labels,
                                                         const int* __restrict__ elens,
                                                         const int* __restrict__ ylens, int blank,
                                                         float* __restrict__ lp, const UttRows ur) {
  const int b = blockIdx.y, t = blockIdx.x;
  const int Sb = 2 * ylens[b] + 1;
  const bool live = t < elens[b];
  const long row = (long)b * Tn + t;                         // row of the lattice tables [B, Tn, S]
  const long lrow = ur.row0 ? ur.row0[b] + t : row;         // row of logits / lse (stacked micro-batches: per utterance)
  const float l = live ? lse[lrow] : 0.f;
  for (int s = threadIdx.x; s < S; s += blockDim.x) {
    float v = -INFINITY;
    if (live && s < Sb) v = to_f32(logits[lrow * ld + ext_label(labels + (long)b * Lmax, s, blank)]) - l;
    lp[row * S + s] = v;
  }
}

// blocks [0,B): alpha (forward in time); blocks [B,2B): beta (backward in time).
__global__ __launch_bounds__(1024) void ctc_lattice_kernel(int B, int Tn, int S, int Lmax,
                                                           const float* __restrict__ lp,
                                                           const int* __restrict__ labels,
                                                           const int* __restrict__ elens,
                                                           const int* __restrict__ ylens, int blank,
                                                           float* __restrict__ alpha, float* __restrict__ beta,
                                                           float* __restrict__ nll) {
  extern __shared__ float sh[];  // [2][S + 4]
  const bool fwd = blockIdx.x < B;
  const int b = fwd ? blockIdx.x : blockIdx.x - B;
  const int s = threadIdx.x;
  const int len = elens[b], L = ylens[b], Sb = 2 * L + 1;
  const int* lab = labels + (long)b * Lmax;
  float* out = (fwd ? alpha : beta) + (long)b * Tn * S;
  const float* lpb = lp + (long)b * Tn * S;
  if (len <= 0) {
    if (fwd && s == 0) nll[b] = L == 0 ? 0.f : INFINITY;
    return;
  }
  // can state s receive the skip transition (from s-2 forward / s+2 backward)?
  bool skip = false;
  const int SP = S + 4;
  float* buf0 = sh + 2;           // index -2..S+1 valid
  float* buf1 = sh + SP + 2;
  if (s < Sb && (s & 1)) {
    const int me = lab[s >> 1];
    if (fwd) skip = s >= 3 && lab[(s >> 1) - 1] != me;
    else skip = s + 2 < Sb && lab[(s >> 1) + 1] != me;
  }
  for (int i = threadIdx.x; i < 2 * SP; i += blockDim.x) sh[i] = -INFINITY;
  __syncthreads();
  // initial frame
  {
    const int t = fwd ? 0 : len - 1;
    float v = -INFINITY;
    if (s < Sb) {
      const bool start = fwd ? (s <= 1) : (s >= Sb - 2);
      if (start) v = lpb[(long)t * S + s];
    }
    if (s < S) { out[(long)t * S + s] = v; buf0[s] = v; }
  }
  __syncthreads();
  float* prev = buf0; float* cur = buf1;
  // The emission log-probabilities do not depend on the recursion: they are fetched CH frames ahead (two register sets, as in the
  // transducer lattice, csrc/rnnt.hip), so a frame costs an LDS exchange + barrier instead of a dependent global load on top
  // (0.7 us per frame before: 224 us for the stacked step's lattices).
  constexpr int CH = 8;
  float eA[CH], eB[CH];
  auto fetch = [&](int i0, float (&e)[CH]) {
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int i = i0 + k;
      const int t = fwd ? i : len - 1 - i;
      e[k] = (i < len && s < Sb) ? lpb[(long)t * S + s] : -INFINITY;
    }
  };
  fetch(1, eA);
  for (int i0 = 1; i0 < len; i0 += CH) {
    fetch(i0 + CH, eB);
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int i = i0 + k;
      if (i >= len) break;   // (block-uniform)
      const int t = fwd ? i : len - 1 - i;
      float v = -INFINITY;
      if (s < Sb) {
        const float a0 = prev[s];
        const float a1 = fwd ? prev[s - 1] : prev[s + 1];
        const float a2 = skip ? (fwd ? prev[s - 2] : prev[s + 2]) : -INFINITY;
        v = lat_add3(a0, a1, a2) + eA[k];
      }
      if (s < S) { out[(long)t * S + s] = v; cur[s] = v; }
      lds_barrier();   // (the alpha / beta rows written to global memory are not read here)
      float* tmp = prev; prev = cur; cur = tmp;
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) eA[k] = eB[k];
  }
  if (fwd && s == 0) {
    const float a = prev[Sb - 1];
    const float c = Sb >= 2 ? prev[Sb - 2] : -INFINITY;
    nll[b] = -log_add(a, c);
  }
}

// one block per (b,t) row: softmax row in LDS, subtract state occupancies, scale, store.
template <typename T>
__global__ __launch_bounds__(256) void ctc_grad_kernel(int Tn, int V, int S, int Lmax,
                                                       const T* __restrict__ logits, long ld,
                                                       const float* __restrict__ lse,
                                                       const int* __restrict__ labels,
                                                       const int* __restrict__ elens,
                                                       const int* __restrict__ ylens, int blank,
                                                       const float* __restrict__ lp,
                                                       const float* __restrict__ alpha,
                                                       const float* __restrict__ beta,
                                                       const float* __restrict__ nll, float gscale,
                                                       const float* __restrict__ gscale_dev,
                                                       T* __restrict__ grad, long ldg, const UttRows ur) {
  extern __shared__ float rowbuf[];  // [V]
  const int b = blockIdx.y, t = blockIdx.x;
  if (ur.tpad && t >= ur.tpad[b]) return;                   // (the grid follows the longest micro-batch)
  const long row = (long)b * Tn + t;                         // row of the lattice tables
  const long lrow = ur.row0 ? ur.row0[b] + t : row;         // row of logits / lse / grad
  T* g = grad + lrow * ldg;
  const float nl = nll[b];
  const bool vec = (V % 8 == 0) && (ld % 8 == 0) && (ldg % 8 == 0);  // 16-byte row accesses
  if (t >= elens[b] || !isfinite(nl)) {
    if (vec) {
      float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int v = threadIdx.x * 8; v < V; v += 2048) store8<T>(g + v, z);
    } else {
      for (int v = threadIdx.x; v < V; v += 256) g[v] = from_f32<T>(0.f);
    }
    return;
  }
  const float l = lse[lrow];
  const T* lg = logits + lrow * ld;
  if (vec) {
    for (int v = threadIdx.x * 8; v < V; v += 2048) {
      float x[8];
      load8<T>(lg + v, x);
#pragma unroll
      for (int e = 0; e < 8; ++e) rowbuf[v + e] = __expf(x[e] - l);
    }
  } else {
    for (int v = threadIdx.x; v < V; v += 256) rowbuf[v] = __expf(to_f32(lg[v]) - l);
  }
  __syncthreads();
  const int Sb = 2 * ylens[b] + 1;
  const int* lab = labels + (long)b * Lmax;
  for (int s = threadIdx.x; s < Sb; s += 256) {
    const long o = row * S + s;
    const float occ = __expf(alpha[o] + beta[o] - lp[o] + nl);
    if (occ > 0.f) atomicAdd(&rowbuf[ext_label(lab, s, blank)], -occ);
  }
  __syncthreads();
  const float gs = (gscale_dev ? gscale * gscale_dev[0] : gscale) * (ur.uscale ? ur.uscale[b] : 1.f);
  if (vec) {
    for (int v = threadIdx.x * 8; v < V; v += 2048) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = rowbuf[v + e] * gs;
      store8<T>(g + v, x);
    }
  } else {
    for (int v = threadIdx.x; v < V; v += 256) g[v] = from_f32<T>(rowbuf[v] * gs);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void argmax_kernel(int V, const T* __restrict__ logits, long ld,
                                                     int* __restrict__ best) {
  __shared__ float rv[4];
  __shared__ int ri[4];
  const T* row = logits + (long)blockIdx.x * ld;
  float m = -INFINITY; int mi = 0x7fffffff;
  for (int v = threadIdx.x; v < V; v += 256) {
    const float x = to_f32(row[v]);
    if (x > m || (x == m && v < mi)) { m = x; mi = v; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o, 64);
    const int oi = __shfl_xor(mi, o, 64);
    if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { rv[threadIdx.x >> 6] = m; ri[threadIdx.x >> 6] = mi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (rv[w] > m || (rv[w] == m && ri[w] < mi)) { m = rv[w]; mi = ri[w]; }
    best[blockIdx.x] = mi == 0x7fffffff ? 0 : mi;
  }
}

// one wave per utterance, 64 frames per pass: keep flags by ballot, output positions by prefix popcount -- the frame order of the
// sequential definition (decoders/ctc.py:176-201: drop repeats, then blanks).  (One THREAD per utterance walked its frames with a
// dependent global load each: 39 us for a 12 s utterance, 2 % of a batch-1 decode.)
__global__ __launch_bounds__(64) void collapse_kernel(int B, int Tn, const int* __restrict__ best, const int* __restrict__ elens,
                                                      int blank, int* __restrict__ hyp, int* __restrict__ hyplen) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= B) return;
  const int len = min(elens[b], Tn);
  int n = 0, carry = -1;
  for (int t0 = 0; t0 < len; t0 += 64) {
    const int t = t0 + lane;
    const int v = t < len ? best[(long)b * Tn + t] : blank;
    int prev = __shfl_up(v, 1, 64);
    if (lane == 0) prev = carry;
    const bool keep = t < len && v != prev && v != blank;
    const unsigned long long mask = __ballot(keep);
    if (keep) hyp[(long)b * Tn + n + __popcll(mask & ((1ull << lane) - 1ull))] = v;
    n += __popcll(mask);
    carry = __shfl(v, 63, 64);
  }
  if (lane == 0) hyplen[b] = n;
}

}  // namespace

extern "C" int emoasr_row_lse(int dtype, int M, int V, const void* logits, long ld, float* lse,
                              void* stream) {
  if (M == 0) return 0;
  const int vec = dtype == EMO_BF16 ? 8 : 4;
  const int aligned = (ld % vec == 0) && (((uintptr_t)logits & 15) == 0);
  EMO_DISPATCH(dtype, (row_lse_kernel<T><<<M, 256, 0, (hipStream_t)stream>>>(V, (const T*)logits, ld, lse,
                                                                            aligned)));
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_ctc_forward(int dtype, int B, int Tn, int V, int Lmax, const void* logits, long ld,
                                  const float* lse, const int* labels, const int* elens, const int* ylens,
                                  int blank, float* lp, float* alpha, float* beta, float* nll,
                                  void* stream) {
  return emoasr_ctc_forward_rows(dtype, B, Tn, V, Lmax, logits, ld, lse, labels, elens, ylens, blank, nullptr, lp, alpha, beta, nll,
                                 stream);
}

extern "C" int emoasr_ctc_forward_rows(int dtype, int B, int Tn, int V, int Lmax, const void* logits, long ld,
                                       const float* lse, const int* labels, const int* elens, const int* ylens,
                                       int blank, const long* row0, float* lp, float* alpha, float* beta, float* nll,
                                       void* stream) {
  const int S = 2 * Lmax + 1;
  const UttRows ur{row0, nullptr, nullptr};
  EMO_CHECK(S <= 1024, "ctc: 2*Lmax+1=%d exceeds 1024 lattice states", S);
  if (B == 0 || Tn == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  dim3 ggrid(Tn, B);
  const int gthreads = std::min(256, cdiv(S, 64) * 64);
  EMO_DISPATCH(dtype, (ctc_gather_kernel<T><<<ggrid, gthreads, 0, s>>>(Tn, S, Lmax, (const T*)logits, ld, lse,
                                                                      labels, elens, ylens, blank, lp, ur)));
  const int threads = cdiv(S, 64) * 64;
  ctc_lattice_kernel<<<2 * B, threads, sizeof(float) * 2 * (S + 4), s>>>(B, Tn, S, Lmax, lp, labels, elens,
                                                                        ylens, blank, alpha, beta, nll);
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_ctc_grad(int dtype, int B, int Tn, int V, int Lmax, const void* logits, long ld,
                               const float* lse, const int* labels, const int* elens, const int* ylens,
                               int blank, const float* lp, const float* alpha, const float* beta,
                               const float* nll, float gscale, const float* gscale_dev, void* grad, long ldg,
                               void* stream) {
  return emoasr_ctc_grad_rows(dtype, B, Tn, V, Lmax, logits, ld, lse, labels, elens, ylens, blank, lp, alpha, beta, nll, gscale,
                              gscale_dev, nullptr, nullptr, nullptr, grad, ldg, stream);
}

extern "C" int emoasr_ctc_grad_rows(int dtype, int B, int Tn, int V, int Lmax, const void* logits, long ld,
                                    const float* lse, const int* labels, const int* elens, const int* ylens,
                                    int blank, const float* lp, const float* alpha, const float* beta,
                                    const float* nll, float gscale, const float* gscale_dev, const long* row0, const int* tpad,
                                    const float* uscale, void* grad, long ldg, void* stream) {
  const int S = 2 * Lmax + 1;
  EMO_CHECK((row0 != nullptr) == (tpad != nullptr), "ctc_grad_rows: row0 and tpad go together");
  const UttRows ur{row0, tpad, uscale};
  EMO_CHECK((size_t)V * 4 <= 160 * 1024 - 256, "ctc_grad: V=%d too large for an LDS row", V);
  if (B == 0 || Tn == 0) return 0;
  dim3 grid(Tn, B);
  EMO_DISPATCH(dtype, {
    if ((size_t)V * 4 > 64 * 1024)
      hipFuncSetAttribute((const void*)ctc_grad_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, V * 4);
    ctc_grad_kernel<T><<<grid, 256, sizeof(float) * V, (hipStream_t)stream>>>(
        Tn, V, S, Lmax, (const T*)logits, ld, lse, labels, elens, ylens, blank, lp, alpha, beta, nll, gscale,
        gscale_dev,
        (T*)grad, ldg, ur);
  });
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_ctc_greedy(int dtype, int B, int Tn, int V, const void* logits, long ld,
                                 const int* elens, int blank, int* best, int* hyp, int* hyplen,
                                 void* stream) {
  if (B == 0 || Tn == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  EMO_DISPATCH(dtype, (argmax_kernel<T><<<B * Tn, 256, 0, s>>>(V, (const T*)logits, ld, best)));
  collapse_kernel<<<B, 64, 0, s>>>(B, Tn, best, elens, blank, hyp, hyplen);
  EMO_LAUNCH_CHECK();
  return 0;
}
