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
