// LayerNorm forward / backward over rows of N features (N <= 1024, N % 4 == 0).
// Reference call sites: nn.LayerNorm in asr/modeling/conformer.py:183-187 (eps 1e-5),
// encoders/transformer.py:73 and transformer.py:140-141,178-180 (eps 1e-12).
//
// HBM-bound: one wave per row, 4 consecutive features per lane per pass, row kept
// in registers between the statistics and the normalisation (x is read once).
#include <algorithm>
#include "common.h"
#include "../../include/emoasr_hip.h"

namespace {

constexpr int LN_MAXC = 4;  // chunks of 256 features -> N <= 1024

template <typename T>
__device__ __forceinline__ void load4(const T* p, float (&o)[4]) {
  if constexpr (sizeof(T) == 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
  } else {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3];
  }
}
template <typename T>
__device__ __forceinline__ void store4(T* p, const float (&o)[4]) {
  if constexpr (sizeof(T) == 4) {
    *reinterpret_cast<f32x4*>(p) = f32x4{o[0], o[1], o[2], o[3]};
  } else {
    bf16x4 v;
    v[0] = (bf16)o[0]; v[1] = (bf16)o[1]; v[2] = (bf16)o[2]; v[3] = (bf16)o[3];
    *reinterpret_cast<bf16x4*>(p) = v;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(int M, int N, const T* __restrict__ x,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps,
                                                     T* __restrict__ y, float* __restrict__ mean,
                                                     float* __restrict__ rstd) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const T* xr = x + (long)row * N;
  float v[LN_MAXC][4];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c) {
    const int col = c * 256 + lane * 4;
    if (col < N) {
      load4(xr + col, v[c]);
      s += v[c][0] + v[c][1] + v[c][2] + v[c][3];
    }
  }
  const float mu = wave_sum(s) / N;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c) {
    const int col = c * 256 + lane * 4;
    if (col < N) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = v[c][j] - mu; q += d * d; }
    }
  }
  const float rs = rsqrtf(wave_sum(q) / N + eps);
  if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
  T* yr = y + (long)row * N;
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c) {
    const int col = c * 256 + lane * 4;
    if (col < N) {
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (v[c][j] - mu) * rs * gamma[col + j] + beta[col + j];
      store4(yr + col, o);
    }
  }
}

// Each block walks ROWS_PER_BLOCK rows (one wave per row at a time) and keeps the
// per-feature dgamma / dbeta partial sums in registers; one LDS reduction and one
// f32 atomic per feature per block at the end.
constexpr int LN_BWD_MAXBLK = 256;  // persistent blocks, 4 rows (one per wave) in flight each

template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(int M, int N, const T* __restrict__ dy,
                                                     const T* __restrict__ x,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ mean,
                                                     const float* __restrict__ rstd,
                                                     const T* __restrict__ dres, T* __restrict__ dx,
                                                     float* __restrict__ dgamma_part) {
  __shared__ float red[4][2][LN_MAXC * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float dg[LN_MAXC][4], db[LN_MAXC][4], g[LN_MAXC][4];
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c) {
    const int col = c * 256 + lane * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dg[c][j] = 0.f; db[c][j] = 0.f;
      g[c][j] = col < N ? gamma[col + j] : 0.f;
    }
  }
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    const float mu = mean[row], rs = rstd[row];
    float xh[LN_MAXC][4], gy[LN_MAXC][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < N) {
        float xv[4], dv[4];
        load4(x + (long)row * N + col, xv);
        load4(dy + (long)row * N + col, dv);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xh[c][j] = (xv[j] - mu) * rs;
          gy[c][j] = dv[j] * g[c][j];
          s1 += gy[c][j];
          s2 += gy[c][j] * xh[c][j];
          dg[c][j] += dv[j] * xh[c][j];
          db[c][j] += dv[j];
        }
      }
    }
    s1 = wave_sum(s1) / N;
    s2 = wave_sum(s2) / N;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < N) {
        float o[4];
        if (dres) load4(dres + (long)row * N + col, o);
        else { o[0] = o[1] = o[2] = o[3] = 0.f; }
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] += rs * (gy[c][j] - s1 - xh[c][j] * s2);
        store4(dx + (long)row * N + col, o);
      }
    }
  }
  if (!dgamma_part) return;
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      red[wave][0][c * 256 + lane * 4 + j] = dg[c][j];
      red[wave][1][c * 256 + lane * 4 + j] = db[c][j];
    }
  __syncthreads();
  // per-block partial sums; a second tiny kernel folds them (no contended atomics)
  float* part = dgamma_part + (long)blockIdx.x * 2 * N;
  for (int col = threadIdx.x; col < N; col += 256) {
    part[col] = red[0][0][col] + red[1][0][col] + red[2][0][col] + red[3][0][col];
    part[N + col] = red[0][1][col] + red[1][1][col] + red[2][1][col] + red[3][1][col];
  }
}

// 64 columns per block, 4 row groups (one per wave) summed through LDS
__global__ __launch_bounds__(256) void ln_bwd_finalize_kernel(int nblk, int N, const float* __restrict__ part,
                                                              float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;  // over 2N
  float s = 0.f;
  if (col < 2 * N) {
#pragma unroll 8
    for (int b = wave; b < nblk; b += 4) s += part[(long)b * 2 * N + col];
  }
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && col < 2 * N) {
    s = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    if (col < N) { if (dgamma) dgamma[col] += s; }
    else if (dbeta) dbeta[col - N] += s;
  }
}

}  // namespace

extern "C" int emoasr_layernorm_fwd(int dtype, int M, int N, const void* x, const float* gamma,
                                    const float* beta, float eps, void* y, float* mean, float* rstd,
                                    void* stream) {
  EMO_CHECK(N % 4 == 0 && N <= LN_MAXC * 256, "layernorm: N=%d unsupported", N);
  if (M == 0) return 0;
  EMO_DISPATCH(dtype, (ln_fwd_kernel<T><<<cdiv(M, 4), 256, 0, (hipStream_t)stream>>>(
                          M, N, (const T*)x, gamma, beta, eps, (T*)y, mean, rstd)));
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_layernorm_bwd(int dtype, int M, int N, const void* dy, const void* x,
                                    const float* gamma, const float* mean, const float* rstd,
                                    const void* dres, void* dx, float* dgamma, float* dbeta,
                                    float* scratch, void* stream) {
  EMO_CHECK(N % 4 == 0 && N <= LN_MAXC * 256, "layernorm: N=%d unsupported", N);
  if (M == 0) return 0;
  const bool want = dgamma || dbeta;
  EMO_CHECK(!want || scratch, "layernorm_bwd: scratch (%d floats) required for dgamma/dbeta",
            LN_BWD_MAXBLK * 2 * N);
  const int nblk = std::min(cdiv(M, 4), LN_BWD_MAXBLK);
  EMO_DISPATCH(dtype, (ln_bwd_kernel<T><<<nblk, 256, 0, (hipStream_t)stream>>>(
                          M, N, (const T*)dy, (const T*)x, gamma, mean, rstd, (const T*)dres, (T*)dx,
                          want ? scratch : nullptr)));
  if (want)
    ln_bwd_finalize_kernel<<<cdiv(2 * N, 64), 256, 0, (hipStream_t)stream>>>(nblk, N, scratch, dgamma, dbeta);
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_layernorm_bwd_scratch_floats(int N) { return LN_BWD_MAXBLK * 2 * N; }
