// Knowledge-distillation kernels:
//
//   emoasr_soft_ce        cross-entropy of log_softmax(logits) against a mix of a dense soft target row
//                         (gathered by index) and a label-smoothed hard target; loss per row + gradient.
//                         One kernel serves DistillLoss (asr/criteria.py:49-100), CTCAlignDistillLoss
//                         (:103-168), RNNTWordDistillLoss (:218-247) and RNNTAlignDistillLoss (:250-288).
//   emoasr_ctc_best_path  the forced-alignment pass of CTCForcedAligner.__call__
//                         (asr/modeling/decoders/ctc_aligner.py:139-221): per frame the arg-max of
//                         alpha*beta over the lattice states reachable from the previous choice.
//   emoasr_ctc_label_map  CTCAlignDistillLoss._frame_to_label_mapping (asr/criteria.py:170-215).
//   emoasr_rnnt_best_path the alignment walk of RNNTForcedAligner.__call__
//                         (asr/modeling/decoders/rnnt_aligner.py:158-198) over the transducer alpha+beta grid.
#include "common.h"
#include "../../include/emoasr_hip.h"

namespace {

// One block per row r.  z = logits[lrow ? lrow[r] : r, :], p = softmax(z), q_s = soft[src[r], :] (f32),
// q_h = label-smoothed one-hot of hard[r] (1-eps on the label, eps/(V-1) elsewhere, criteria.py:9-15).
//   loss[r]   = -( ws[r] * sum_v q_s[v] log p[v]  +  wh[r] * sum_v q_h[v] log p[v] )
//   grad[., v] = g * ( ws * (p[v] * sum(q_s) - q_s[v]) + wh * (p[v] - q_h[v]) )
// src[r] < 0 drops the soft term, hard[r] < 0 the hard term; a row with neither gets loss 0 / zero gradient.
template <typename T>
__global__ __launch_bounds__(256) void soft_ce_kernel(int V, const T* __restrict__ logits, long ld,
                                                      const int* __restrict__ lrow,
                                                      const float* __restrict__ soft, long lds,
                                                      const int* __restrict__ src,
                                                      const int* __restrict__ hard,
                                                      const float* __restrict__ ws,
                                                      const float* __restrict__ wh, float eps,
                                                      float* __restrict__ loss, float gscale,
                                                      const float* __restrict__ gscale_dev,
                                                      T* __restrict__ grad, long ldg) {
  __shared__ float red[16];
  const long r = blockIdx.x;
  const long lr = lrow ? lrow[r] : r;
  const T* row = logits + lr * ld;
  const int sr = src ? src[r] : -1;
  const int y = hard ? hard[r] : -1;
  const float w_s = (sr >= 0 && ws) ? ws[r] : 0.f;
  const float w_h = (y >= 0 && wh) ? wh[r] : 0.f;
  if (w_s == 0.f && w_h == 0.f) {
    if (threadIdx.x == 0) loss[r] = 0.f;
    if (grad) for (int v = threadIdx.x; v < V; v += 256) grad[lr * ldg + v] = from_f32<T>(0.f);
    return;
  }
  const float* q = soft + (long)(sr < 0 ? 0 : sr) * lds;
  const bool use_s = w_s != 0.f;
  float mx = -INFINITY, sz = 0.f, sq = 0.f, sqz = 0.f;
  for (int v = threadIdx.x; v < V; v += 256) {
    const float x = to_f32(row[v]);
    mx = fmaxf(mx, x);
    sz += x;
    if (use_s) { const float qv = q[v]; sq += qv; sqz += qv * x; }
  }
  mx = block_max(mx, red);
  sz = block_sum(sz, red);
  sq = block_sum(sq, red);
  sqz = block_sum(sqz, red);
  float se = 0.f;
  for (int v = threadIdx.x; v < V; v += 256) se += __expf(to_f32(row[v]) - mx);
  se = block_sum(se, red);
  const float lse = mx + logf(se);
  const float off = eps / (V - 1);
  if (threadIdx.x == 0) {
    float l = 0.f;
    if (use_s) l += w_s * (sqz - lse * sq);
    if (w_h != 0.f) {
      const float lpy = to_f32(row[y]) - lse;
      l += w_h * ((1.f - eps) * lpy + off * (sz - V * lse - lpy));
    }
    loss[r] = -l;
  }
  if (grad) {
    const float g = gscale_dev ? gscale * gscale_dev[0] : gscale;
    const float cp = g * (w_s * sq + w_h);
    for (int v = threadIdx.x; v < V; v += 256) {
      const float pv = __expf(to_f32(row[v]) - lse);
      float d = cp * pv;
      if (use_s) d -= g * w_s * q[v];
      if (w_h != 0.f) d -= g * w_h * (v == y ? 1.f - eps : off);
      grad[lr * ldg + v] = from_f32<T>(d);
    }
  }
}

// One block per utterance, one thread per lattice state.  post[t,s] = alpha[t,s] + beta[t,s] - lp[t,s]
// (alpha, beta both include the emission at t, so this is alpha_t * beta_t / y_t, the quantity the
// reference accumulates in log_probs_fwd_bwd, ctc_aligner.py:178-194).  Frame 0 may start in states
// {0,1}; from state o the next frame may take {o, o+1, o+2 unless ext[o+2] == ext[o]} (:118-124).
// arg-max ties -> the lowest state (torch.argmax).  aligns[b,t] = ext label of the chosen state for
// t < elens[b], 0 beyond (the reference's zero-initialised best_aligns, :196).
__global__ __launch_bounds__(1024) void ctc_best_path_kernel(int Tn, int S, int Lmax,
                                                             const float* __restrict__ lp,
                                                             const float* __restrict__ alpha,
                                                             const float* __restrict__ beta,
                                                             const int* __restrict__ labels,
                                                             const int* __restrict__ elens,
                                                             const int* __restrict__ ylens, int blank,
                                                             int* __restrict__ aligns) {
  extern __shared__ float post[];  // [2][S + 2]
  __shared__ int cur;
  const int b = blockIdx.x, s = threadIdx.x;
  const int len = min(elens[b], Tn), L = ylens[b], Sb = 2 * L + 1;
  const int* lab = labels + (long)b * Lmax;
  const long base = (long)b * Tn * S;
  int* out = aligns + (long)b * Tn;
  for (int t = len + s; t < Tn; t += blockDim.x) out[t] = 0;
  if (len <= 0) return;
  const int SP = S + 2;
  auto fetch = [&](int t) -> float {
    if (s >= Sb) return -INFINITY;
    const long i = base + (long)t * S + s;
    const float e = lp[i];
    return e == -INFINITY ? -INFINITY : alpha[i] + beta[i] - e;
  };
  float nxt = fetch(0);
  if (s == 0) cur = -1;
  for (int t = 0; t < len; ++t) {
    float* buf = post + (t & 1) * SP;
    if (s < S) buf[s] = nxt;
    if (s == 0) { buf[S] = -INFINITY; buf[S + 1] = -INFINITY; }
    if (t + 1 < len) nxt = fetch(t + 1);  // in flight while thread 0 decides
    __syncthreads();
    if (s == 0) {
      const int o = cur;
      int best;
      if (o < 0) {
        best = (Sb > 1 && buf[1] > buf[0]) ? 1 : 0;
      } else {
        best = o;
        float bv = buf[o];
        if (o + 1 < Sb && buf[o + 1] > bv) { best = o + 1; bv = buf[o + 1]; }
        if ((o & 1) && o + 2 < Sb && lab[(o + 2) >> 1] != lab[o >> 1] && buf[o + 2] > bv) best = o + 2;
      }
      cur = best;
      out[t] = (best & 1) ? lab[best >> 1] : blank;
    }
    // the next iteration writes the other buffer; thread 0's reads of this one finish before it reaches
    // the barrier of iteration t+1, and buffer (t&1) is rewritten only at iteration t+2
  }
}

// One block per utterance.  A frame starts a new label when its token is not blank and differs from the
// previous frame's token (criteria.py:178-181); label ids count those starts.  position: 0 all frames of
// the label, 1 first frame, 2 middle frame ((left+right)//2), 3 last frame.  map = -1 elsewhere.
// count[b] = number of frames with map >= 0.
__global__ __launch_bounds__(256) void ctc_label_map_kernel(int Tn, const int* __restrict__ aligns,
                                                            const int* __restrict__ xlens, int blank,
                                                            int position, int* __restrict__ map,
                                                            int* __restrict__ count) {
  __shared__ int wsum[4];
  __shared__ int carry;
  __shared__ int total;
  const int b = blockIdx.x;
  const int len = min(xlens[b], Tn);
  const int* al = aligns + (long)b * Tn;
  int* mp = map + (long)b * Tn;
  if (threadIdx.x == 0) { carry = 0; total = 0; }
  __syncthreads();
  for (int t0 = 0; t0 < Tn; t0 += 256) {
    const int t = t0 + threadIdx.x;
    const int tok = t < len ? al[t] : blank;
    const int prev = (t > 0 && t < len) ? al[t - 1] : -1;
    const int start = (t < len && tok != blank && (t == 0 || tok != prev)) ? 1 : 0;
    // inclusive scan of `start` over the 256 frames of this chunk
    int x = start;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o, 64);
      if ((threadIdx.x & 63) >= o) x += y;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
    __syncthreads();
    int add = carry;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) add += wsum[w];
    const int id = x + add - 1;  // label index of this frame's token (if not blank)
    int m = -1;
    if (t < len && tok != blank) {
      if (position == 0) {
        m = id;
      } else {
        // run extents: same non-blank token on consecutive frames
        int l = t, r = t;
        while (l > 0 && al[l - 1] == tok) --l;
        while (r + 1 < len && al[r + 1] == tok) ++r;
        const int pick = position == 1 ? l : (position == 3 ? r : (l + r) / 2);
        if (t == pick) m = id;
      }
    }
    if (t < Tn) mp[t] = m;
    int c = m >= 0 ? 1 : 0;
    c = (int)wave_sum((float)c);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) atomicAdd(&total, c);
    if (threadIdx.x == 255) carry = x + add;
    __syncthreads();
  }
  if (threadIdx.x == 0) count[b] = total;
}

// One wave per utterance; lane 0 walks the lattice (rnnt_aligner.py:186-196): from (0,0), while t+1 < T and
// u < U: move to the next frame if alpha+beta is larger there than one label up, else emit label u at frame t.
// Labels not emitted before the last frame keep frame 0 (the reference's zero-initialised best_aligns).
__global__ __launch_bounds__(64) void rnnt_best_path_kernel(int Tn, int U, const float* __restrict__ alpha,
                                                            const float* __restrict__ beta,
                                                            const int* __restrict__ elens,
                                                            const int* __restrict__ ylens, int* __restrict__ aligns) {
  const int b = blockIdx.x;
  int* out = aligns + (long)b * (U - 1);
  for (int i = threadIdx.x; i < U - 1; i += 64) out[i] = 0;
  __syncthreads();
  if (threadIdx.x != 0) return;
  const int T = min(elens[b], Tn), Ub = min(ylens[b], U - 1);
  const float* a = alpha + (long)b * Tn * U;
  const float* bt = beta + (long)b * Tn * U;
  int t = 0, u = 0;
  while (t + 1 < T && u < Ub) {
    const long down = (long)(t + 1) * U + u, right = (long)t * U + u + 1;
    if (a[down] + bt[down] > a[right] + bt[right]) ++t;
    else { out[u] = t; ++u; }
  }
}

}  // namespace

extern "C" int emoasr_rnnt_best_path(int B, int Tn, int U, const float* alpha, const float* beta, const int* elens,
                                     const int* ylens, int* aligns, void* stream) {
  if (B == 0 || U <= 1) return 0;
  rnnt_best_path_kernel<<<B, 64, 0, (hipStream_t)stream>>>(Tn, U, alpha, beta, elens, ylens, aligns);
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_soft_ce(int dtype, int R, int V, const void* logits, long ld, const int* lrow,
                              const float* soft, long lds, const int* src, const int* hard, const float* w_soft,
                              const float* w_hard, float lsm_prob, float* loss, float gscale,
                              const float* gscale_dev, void* grad, long ldg, void* stream) {
  if (R == 0) return 0;
  EMO_CHECK(V >= 2, "soft_ce: V=%d", V);
  EMO_CHECK(soft || !src, "soft_ce: row indices without a soft-label table");
  EMO_DISPATCH(dtype, (soft_ce_kernel<T><<<R, 256, 0, (hipStream_t)stream>>>(
                          V, (const T*)logits, ld, lrow, soft, lds, src, hard, w_soft, w_hard, lsm_prob, loss, gscale,
                          gscale_dev, (T*)grad, ldg)));
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_ctc_best_path(int B, int Tn, int Lmax, const float* lp, const float* alpha, const float* beta,
                                    const int* labels, const int* elens, const int* ylens, int blank, int* aligns,
                                    void* stream) {
  const int S = 2 * Lmax + 1;
  EMO_CHECK(S <= 1024, "ctc_best_path: 2*Lmax+1=%d exceeds 1024 lattice states", S);
  if (B == 0 || Tn == 0) return 0;
  const int threads = cdiv(S, 64) * 64;
  ctc_best_path_kernel<<<B, threads, sizeof(float) * 2 * (S + 2), (hipStream_t)stream>>>(
      Tn, S, Lmax, lp, alpha, beta, labels, elens, ylens, blank, aligns);
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_ctc_label_map(int B, int Tn, const int* aligns, const int* xlens, int blank, int position,
                                    int* label_map, int* count, void* stream) {
  EMO_CHECK(position >= 0 && position <= 3, "ctc_label_map: position=%d", position);
  if (B == 0) return 0;
  ctc_label_map_kernel<<<B, 256, 0, (hipStream_t)stream>>>(Tn, aligns, xlens, blank, position, label_map, count);
  EMO_LAUNCH_CHECK();
  return 0;
}
