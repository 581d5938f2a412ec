// Optimizer-side kernels on flat f32 buffers: squared gradient norm, Adam.
// Reference: asr/train_asr.py:84-92 (clip_grad_norm_, NaN skip) + torch.optim.Adam with
// coupled L2 weight decay (train_asr.py:228), lr set by ScheduledOptimizer (optimizers.py:56-82).
#include <algorithm>
#include "common.h"
#include "../../include/emoasr_hip.h"

namespace {

// Bit-reproducible: every block leaves its partial in `part`, the block that arrives last (a ticket counter) adds them up in block
// order.  Data-parallel replicas clip by this norm: with per-block float atomics on `out` two ranks holding the SAME reduced
// gradient got norms that differed in the last bits, and their parameters drifted apart (tests/test_00_dp_two_process_gpu.py).
__global__ __launch_bounds__(256) void sqnorm_kernel(long n, const float* __restrict__ x, float* out, float* __restrict__ part,
                                                     unsigned* __restrict__ ticket) {
  __shared__ float red[16];
  __shared__ int s_last;
  float s = 0.f;
  const long n4 = n / 4;
  // four independent 16-byte loads in flight per thread (one per iteration left the kernel latency-bound
  // at ~1.5 TB/s)
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const f32x4 a = reinterpret_cast<const f32x4*>(x)[i], b = reinterpret_cast<const f32x4*>(x)[i + stride];
    const f32x4 c = reinterpret_cast<const f32x4*>(x)[i + 2 * stride], d = reinterpret_cast<const f32x4*>(x)[i + 3 * stride];
    s += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3] + b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3];
    s += c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3] + d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
  }
  for (; i < n4; i += stride) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0)
    for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) s += x[i] * x[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = s;
    __threadfence();
    s_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  float t = 0.f;
  for (unsigned b = threadIdx.x; b < gridDim.x; b += 256) t += __hip_atomic_load(&part[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();   // (red is reused)
  t = block_sum(t, red);
  if (threadIdx.x == 0) {
    *out += t;
    *ticket = 0u;
  }
}

__global__ __launch_bounds__(256) void adam_kernel(long n, float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, float lr,
                                                   float beta1, float beta2, float eps, float wd, float bc1,
                                                   float bc2_sqrt, const float* __restrict__ gnorm_sq,
                                                   float clip, float grad_mult, int* __restrict__ skipped) {
  float mult = grad_mult;
  if (gnorm_sq) {
    const float nsq = *gnorm_sq * grad_mult * grad_mult;
    if (!isfinite(nsq)) {  // NaN / Inf gradient: skip the step (train_asr.py:88-89) and count it for the host's counters
      if (skipped && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(skipped, 1);
      return;
    }
    if (clip > 0.f) {
      const float c = clip / (sqrtf(nsq) + 1e-6f);
      if (c < 1.f) mult *= c;
    }
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float pi = p[i];
    const float gi = g[i] * mult + wd * pi;
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
  }
}

inline int ew_grid(long n) { long b = (n + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

}  // namespace

extern "C" int emoasr_sqnorm(long n, const float* x, float* out, void* stream) {
  if (n == 0) return 0;
  EMO_CHECK(((uintptr_t)x & 15) == 0, "sqnorm: buffer must be 16-byte aligned");
  // at most 1024 blocks, each leaving one partial sum + the arrival counter: one area per (device, stream) -- calls on one stream are
  // ordered, calls on two streams (two engines, a data-parallel rehearsal on one GPU) no longer share the partials
  EmoScratch* sc = emo_stream_scratch(EMO_SCRATCH_SQNORM, stream, 1025 * sizeof(float));
  if (!sc) return 1;
  float* scratch = static_cast<float*>(sc->dev);
  sqnorm_kernel<<<std::min(ew_grid(n / 4 + 1), 1024), 256, 0, (hipStream_t)stream>>>(n, x, out, scratch,
                                                                                     reinterpret_cast<unsigned*>(scratch + 1024));
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_adam_step_ex(long n, float* p, const float* g, float* m, float* v, float lr, float beta1,
                                   float beta2, float eps, float weight_decay, int step, const float* gnorm_sq, float clip,
                                   float grad_mult, int* skipped, void* stream);
extern "C" int emoasr_adam_step(long n, float* p, const float* g, float* m, float* v, float lr, float beta1,
                                float beta2, float eps, float weight_decay, int step,
                                const float* gnorm_sq, float clip, float grad_mult, void* stream) {
  return emoasr_adam_step_ex(n, p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, gnorm_sq, clip, grad_mult, nullptr, stream);
}

extern "C" int emoasr_adam_step_ex(long n, float* p, const float* g, float* m, float* v, float lr, float beta1,
                                   float beta2, float eps, float weight_decay, int step, const float* gnorm_sq, float clip,
                                   float grad_mult, int* skipped, void* stream) {
  if (n == 0) return 0;
  EMO_CHECK(step >= 1, "adam: step must be >= 1");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2 = 1.f - powf(beta2, (float)step);
  adam_kernel<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(n, p, g, m, v, lr, beta1, beta2, eps, weight_decay,
                                                          bc1, sqrtf(bc2), gnorm_sq, clip, grad_mult, skipped);
  EMO_LAUNCH_CHECK();
  return 0;
}
