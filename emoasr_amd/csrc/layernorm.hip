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
int g_ln_fwd8 = 1;          // option "ln_fwd8": the half-wave-per-row forward kernel for M > 4096 rows
int g_ln_bwd_pf = 1;        // option "ln_bwd_pf": the backward requests the next row before it reduces the current one
int g_ln_bwd_blocks = 512;  // option "ln_bwd_blocks": persistent blocks of the N % 8 == 0 backward (<= LN_BWD8_MAXBLK)

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

// EARLY: gamma / beta requested together with the row (few rows -- decoding: one round trip less, 4.8 -> 4.0 us); otherwise they
// are loaded where they are used (many rows: fewer registers in flight measured faster, 2.35 vs 2.39 ms per training step)
template <typename T, bool EARLY>
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
  float gm[LN_MAXC][4], bt[LN_MAXC][4];   // requested together with the row: a second round trip after the statistics otherwise
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c) {
    const int col = c * 256 + lane * 4;
    if (col < N) {
      load4(xr + col, v[c]);
      if constexpr (EARLY) { load4(gamma + col, gm[c]); load4(beta + col, bt[c]); }
    }
  }
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c) {
    const int col = c * 256 + lane * 4;
    if (col < N) s += v[c][0] + v[c][1] + v[c][2] + v[c][3];
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
      if constexpr (!EARLY) { load4(gamma + col, gm[c]); load4(beta + col, bt[c]); }
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (v[c][j] - mu) * rs * gm[c][j] + bt[c][j];
      store4(yr + col, o);
    }
  }
}

// Each block walks ROWS_PER_BLOCK rows (one wave per row at a time) and keeps the
// per-feature dgamma / dbeta partial sums in registers; one LDS reduction and one
// f32 atomic per feature per block at the end.
constexpr int LN_BWD_MAXBLK = 256;   // general kernel: persistent blocks, 4 rows (one per wave) in flight each
constexpr int LN_BWD8_MAXBLK = 2048;  // N % 8 == 0 kernel: 8 rows (one per half-wave) in flight each; 1024 blocks: 9.3 us + a 26 us fold per step, 512: 8.5 + 15, 256: 9.9 + 8

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

// ---- fast path (N % 8 == 0): one HALF-wave per row, 8 consecutive features (16 bytes of bf16) per
// lane per 256-feature chunk, 8 rows in flight per block and one row per half-wave per sweep at the
// model's sizes (M = 7200 rows -> 900 blocks): the kernel is one round trip of loads deep instead of
// seven dependent ones.  NC = number of 256-feature chunks.
template <typename T>
__device__ __forceinline__ void ln_load8(const T* p, float (&o)[8]) {
  const Vec16<T> v = load16(p);
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = v.get(j);
  } else {
    const Vec16<T> w = load16(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = v.get(j); o[4 + j] = w.get(j); }
  }
}
template <typename T>
__device__ __forceinline__ void ln_store8(T* p, const float (&o)[8]) {
  if constexpr (sizeof(T) == 2) {
    Vec16<T> v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v.set(j, o[j]);
    store16(p, v);
  } else {
    Vec16<T> v, w;
#pragma unroll
    for (int j = 0; j < 4; ++j) { v.set(j, o[j]); w.set(j, o[4 + j]); }
    store16(p, v);
    store16(p + 4, w);
  }
}
// Sum over the 32 lanes of a half wave, in every lane, without the LDS pipeline (__shfl_xor is a ds_bpermute: the five-step trees
// were ten LDS round trips per row in the backward): four DPP adds inside the rows of 16 (quad swaps, half mirror, mirror), then
// v_permlane16_swap hands every lane its partner row's sum.  Sources stay inside the half wave: a half that has left the loop
// (exec off) is never read.
template <int CTRL> __device__ __forceinline__ float ln_dpp(const float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float half_wave_sum(float v) {
  v += ln_dpp<0xB1>(v);    // quad_perm [1, 0, 3, 2]
  v += ln_dpp<0x4E>(v);    // quad_perm [2, 3, 0, 1]
  v += ln_dpp<0x141>(v);   // row_half_mirror
  v += ln_dpp<0x140>(v);   // row_mirror: every lane of a row of 16 holds the row's sum
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);   // (own row, partner row) in either order
}

// 8 consecutive features as they come from memory (16 bytes of bf16, 32 of f32): kept raw across a loop iteration so that the
// NEXT row's loads are in flight while the current row is reduced (converting at the load would wait for it there)
template <typename T> struct LnRaw8 {
  Vec16<T> a[sizeof(T) == 2 ? 1 : 2];
  __device__ __forceinline__ void load(const T* p) {
    a[0] = load16(p);
    if constexpr (sizeof(T) == 4) a[1] = load16(p + 4);
  }
  __device__ __forceinline__ void get(float (&o)[8]) const {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = a[0].get(j);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) { o[j] = a[0].get(j); o[4 + j] = a[1].get(j); }
    }
  }
};

// Forward fast path for many rows (training: M = 35 k stacked rows): one HALF-wave per row, 16 bytes per lane and 256-feature
// chunk -- half as many waves in flight for the same bytes as the 4-features-per-lane kernel, whose 8-byte requests ran the pass at
// ~2.7 TB/s (read + write).  Same two-pass statistics on the registers.
template <typename T, int NC>
__global__ __launch_bounds__(256) void ln_fwd8_kernel(int M, int N, const T* __restrict__ x,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps,
                                                      T* __restrict__ y, float* __restrict__ mean,
                                                      float* __restrict__ rstd) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane & 31, half = lane >> 5;
  const int row = (blockIdx.x * 4 + wave) * 2 + half;
  if (row >= M) return;   // a whole half leaves: the xor offsets of half_wave_sum stay inside the other one
  const T* xr = x + (long)row * N;
  float v[NC][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = c * 256 + hl * 8;
    if (col < N) ln_load8(xr + col, v[c]);
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = c * 256 + hl * 8;
    if (col < N) {
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[c][j];
    }
  }
  const float mu = half_wave_sum(s) / N;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = c * 256 + hl * 8;
    if (col < N) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[c][j] - mu; q += d * d; }
    }
  }
  const float rs = rsqrtf(half_wave_sum(q) / N + eps);
  if (hl == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
  T* yr = y + (long)row * N;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = c * 256 + hl * 8;
    if (col < N) {
      float o[8];
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + col), g1 = *reinterpret_cast<const f32x4*>(gamma + col + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + col), b1 = *reinterpret_cast<const f32x4*>(beta + col + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = (v[c][j] - mu) * rs * g0[j] + b0[j];
        o[4 + j] = (v[c][4 + j] - mu) * rs * g1[j] + b1[j];
      }
      ln_store8(yr + col, o);
    }
  }
}

template <typename T, int NC, bool PF>
__global__ __launch_bounds__(256) void ln_bwd8_kernel(int M, int N, const T* __restrict__ dy,
                                                      const T* __restrict__ x,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ mean,
                                                      const float* __restrict__ rstd,
                                                      const T* __restrict__ dres, T* __restrict__ dx,
                                                      float* __restrict__ dgamma_part, T* __restrict__ dy2,
                                                      float scale2, float p2, uint64_t seed2) {
  extern __shared__ float red8[];  // [4 waves][2][N]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane & 31, half = lane >> 5;
  float dg[NC][8], db[NC][8], g[NC][8];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = c * 256 + hl * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      dg[c][j] = 0.f; db[c][j] = 0.f;
      g[c][j] = col < N ? gamma[col + j] : 0.f;
    }
  }
  // the next row's pieces are requested before the current row is reduced: a block walks M / (8 * gridDim.x) ~ 9 rows per
  // half-wave at the stacked size, each a dependent round trip otherwise
  LnRaw8<T> nx[NC], nd[NC], nr[NC];
  float nmu = 0.f, nrs = 0.f;
  auto fetch = [&](int row) __attribute__((always_inline)) {
    if (row < M) {
      nmu = mean[row]; nrs = rstd[row];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int col = c * 256 + hl * 8;
        if (col < N) {
          nx[c].load(x + (long)row * N + col);
          nd[c].load(dy + (long)row * N + col);
          if (dres) nr[c].load(dres + (long)row * N + col);
        }
      }
    }
  };
  fetch((blockIdx.x * 4 + wave) * 2 + half);
  for (int row = (blockIdx.x * 4 + wave) * 2 + half; row < M; row += gridDim.x * 8) {
    const float mu = nmu, rs = nrs;
    float xh[NC][8], gy[NC][8], o[NC][8];
    float xv_[NC][8], dv_[NC][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = c * 256 + hl * 8;
      if (col < N) {
        nx[c].get(xv_[c]); nd[c].get(dv_[c]);
        if (dres) nr[c].get(o[c]);
        else {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[c][j] = 0.f;
        }
      }
    }
    if (PF) fetch(row + gridDim.x * 8);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = c * 256 + hl * 8;
      if (col < N) {
        const float (&xv)[8] = xv_[c];
        const float (&dv)[8] = dv_[c];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          xh[c][j] = (xv[j] - mu) * rs;
          gy[c][j] = dv[j] * g[c][j];
          s1 += gy[c][j];
          s2 += gy[c][j] * xh[c][j];
          dg[c][j] += dv[j] * xh[c][j];
          db[c][j] += dv[j];
        }
      }
    }
    // the two halves of a wave may run different trip counts; xor offsets < 32 stay inside a half
    s1 = half_wave_sum(s1) / N;
    s2 = half_wave_sum(s2) / N;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = c * 256 + hl * 8;
      if (col < N) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[c][j] += rs * (gy[c][j] - s1 - xh[c][j] * s2);
        ln_store8(dx + (long)row * N + col, o[c]);
        if (dy2) {  // gradient entering the next residual branch x + scale2 * dropout(f(x)), from the ROUNDED dx
          float o2[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) o2[j] = to_f32(from_f32<T>(o[c][j])) * scale2;
          dropout_apply8(seed2, (uint64_t)((long)row * N + col), p2, o2);   // (N % 8 == 0, col % 8 == 0: an even first index)
          ln_store8(dy2 + (long)row * N + col, o2);
        }
      }
    }
    if (!PF) fetch(row + gridDim.x * 8);
  }
  if (!dgamma_part) return;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = c * 256 + hl * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a = dg[c][j] + __shfl_xor(dg[c][j], 32, 64);
      const float b = db[c][j] + __shfl_xor(db[c][j], 32, 64);
      if (half == 0 && col < N) {
        red8[(wave * 2 + 0) * N + col + j] = a;
        red8[(wave * 2 + 1) * N + col + j] = b;
      }
    }
  }
  __syncthreads();
  float* part = dgamma_part + (long)blockIdx.x * 2 * N;
  for (int i = threadIdx.x; i < 2 * N; i += 256) {
    const int which = i / N, col = i - which * N;
    part[i] = red8[(0 * 2 + which) * N + col] + red8[(1 * 2 + which) * N + col] +
              red8[(2 * 2 + which) * N + col] + red8[(3 * 2 + which) * N + col];
  }
}

// Folds the per-block partials: 64 columns (of the 2N) per block in x, a slice of the partial rows
// per block in y (4 waves each), then one f32 atomic per column per block (gridDim.y-way contention
// only; dgamma / dbeta are accumulated into, as torch's .grad is).
__global__ __launch_bounds__(256) void ln_bwd_finalize_kernel(int nblk, int N, const float* __restrict__ part,
                                                              float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;  // over 2N
  const int per = (nblk + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
  float s = 0.f;
  if (col < 2 * N) {
#pragma unroll 8
    for (int b = b0 + wave; b < b1; b += 4) s += part[(long)b * 2 * N + col];
  }
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && col < 2 * N) {
    s = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    if (col < N) { if (dgamma) atomicAdd(&dgamma[col], s); }
    else if (dbeta) atomicAdd(&dbeta[col - N], s);
  }
}

// several deferred finalizations in one launch: blockIdx.z = item
struct LnFinGroup {
  int n;
  int nblk[EMOASR_LN_FINALIZE_MAX];
  emoasr_ln_finalize_item_t it[EMOASR_LN_FINALIZE_MAX];
};
__global__ __launch_bounds__(256) void ln_bwd_finalize_grouped_kernel(const LnFinGroup G) {
  __shared__ float red[4][64];
  const emoasr_ln_finalize_item_t& it = G.it[blockIdx.z];
  const int N = it.N, nblk = G.nblk[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;  // over 2N
  if (blockIdx.x * 64 >= 2 * N) return;
  const int per = (nblk + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
  float s = 0.f;
  if (col < 2 * N) {
#pragma unroll 8
    for (int b = b0 + wave; b < b1; b += 4) s += it.part[(long)b * 2 * N + col];
  }
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && col < 2 * N) {
    s = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    if (col < N) { if (it.dgamma) atomicAdd(&it.dgamma[col], s); }
    else if (it.dbeta) atomicAdd(&it.dbeta[col - N], s);
  }
}

int ln_bwd_nblk(int M, int N) {
  return N % 8 == 0 ? std::min(cdiv(M, 8), g_ln_bwd_blocks) : std::min(cdiv(M, 4), LN_BWD_MAXBLK);
}

}  // namespace

void emo_ln_set_fwd8(int v) { g_ln_fwd8 = v ? 1 : 0; }
void emo_ln_set_bwd_pf(int v) { g_ln_bwd_pf = v ? 1 : 0; }
void emo_ln_set_bwd_blocks(int v) { g_ln_bwd_blocks = std::max(64, std::min(v, LN_BWD8_MAXBLK)); }

extern "C" int emoasr_layernorm_fwd(int dtype, int M, int N, const void* x, const float* gamma,
                                    const float* beta, float eps, void* y, float* mean, float* rstd,
                                    void* stream) {
  EMO_CHECK(N % 4 == 0 && N <= LN_MAXC * 256, "layernorm: N=%d unsupported", N);
  if (M == 0) return 0;
  EmoTimerScope timer_(EMO_TIMER_LAYERNORM, (hipStream_t)stream, 0.0, 2.0 * M * N * (dtype == EMO_BF16 ? 2.0 : 4.0));
  if (M <= 4096) {
    EMO_DISPATCH(dtype, (ln_fwd_kernel<T, true><<<cdiv(M, 4), 256, 0, (hipStream_t)stream>>>(
                            M, N, (const T*)x, gamma, beta, eps, (T*)y, mean, rstd)));
  } else if (N % 8 == 0 && g_ln_fwd8) {
#define EMO_LNF8(NC_)                                                                                      \
  EMO_DISPATCH(dtype, (ln_fwd8_kernel<T, NC_><<<cdiv(M, 8), 256, 0, (hipStream_t)stream>>>(M, N, (const T*)x, gamma, beta, \
                                                                                           eps, (T*)y, mean, rstd)))
    switch (cdiv(N, 256)) {
      case 1: EMO_LNF8(1); break;
      case 2: EMO_LNF8(2); break;
      case 3: EMO_LNF8(3); break;
      default: EMO_LNF8(4); break;
    }
#undef EMO_LNF8
  } else {
    EMO_DISPATCH(dtype, (ln_fwd_kernel<T, false><<<cdiv(M, 4), 256, 0, (hipStream_t)stream>>>(
                            M, N, (const T*)x, gamma, beta, eps, (T*)y, mean, rstd)));
  }
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_layernorm_bwd_ex(int dtype, int M, int N, const void* dy, const void* x,
                                       const float* gamma, const float* mean, const float* rstd,
                                       const void* dres, void* dx, float* dgamma, float* dbeta,
                                       float* scratch, const emoasr_ln_bwd_opts_t* opts, void* stream) {
  EMO_CHECK(N % 4 == 0 && N <= LN_MAXC * 256, "layernorm: N=%d unsupported", N);
  if (M == 0) return 0;
  const bool want = dgamma || dbeta;
  EMO_CHECK(!want || scratch, "layernorm_bwd: scratch (%d floats) required for dgamma/dbeta",
            LN_BWD8_MAXBLK * 2 * N);
  const int nblk = ln_bwd_nblk(M, N);
  hipStream_t s = (hipStream_t)stream;
  // algorithmic bytes: dy, x in; dx out; + the residual gradient in and the dropped-out copy out when present
  EmoTimerScope timer_(EMO_TIMER_LAYERNORM, s, 0.0,
                       (3.0 + (dres ? 1 : 0) + (opts && opts->dy2 ? 1 : 0)) * M * N * (dtype == EMO_BF16 ? 2.0 : 4.0));
  float* part = want ? scratch : nullptr;
  void* dy2 = opts ? opts->dy2 : nullptr;
  const float scale2 = opts ? opts->scale2 : 1.f, p2 = opts ? opts->drop_p2 : 0.f;
  const uint64_t seed2 = opts ? opts->seed2 : 0;
  if (N % 8 == 0) {
    const int smem = 4 * 2 * N * (int)sizeof(float);
#define EMO_LN8_(NC_, PF_)                                                                                     \
  EMO_DISPATCH(dtype, (ln_bwd8_kernel<T, NC_, PF_><<<nblk, 256, smem, s>>>(M, N, (const T*)dy, (const T*)x, gamma, mean, \
                                                                           rstd, (const T*)dres, (T*)dx, part,          \
                                                                           (T*)dy2, scale2, p2, seed2)))
#define EMO_LN8(NC_) do { if (g_ln_bwd_pf && nblk * 8 < M) EMO_LN8_(NC_, true); else EMO_LN8_(NC_, false); } while (0)
    switch (cdiv(N, 256)) {
      case 1: EMO_LN8(1); break;
      case 2: EMO_LN8(2); break;
      case 3: EMO_LN8(3); break;
      default: EMO_LN8(4); break;
    }
#undef EMO_LN8
#undef EMO_LN8_
  } else {
    EMO_DISPATCH(dtype, (ln_bwd_kernel<T><<<nblk, 256, 0, s>>>(M, N, (const T*)dy, (const T*)x, gamma, mean, rstd,
                                                               (const T*)dres, (T*)dx, part)));
    if (dy2 && emoasr_scale_dropout(dtype, (long)M * N, dx, dy2, scale2, p2, seed2, stream)) return 1;
  }
  if (want && !(opts && opts->defer_finalize)) {
    dim3 grid(cdiv(2 * N, 64), std::max(1, std::min(8, nblk / 32)));
    ln_bwd_finalize_kernel<<<grid, 256, 0, s>>>(nblk, N, scratch, dgamma, dbeta);
  }
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_layernorm_bwd(int dtype, int M, int N, const void* dy, const void* x,
                                    const float* gamma, const float* mean, const float* rstd,
                                    const void* dres, void* dx, float* dgamma, float* dbeta,
                                    float* scratch, void* stream) {
  return emoasr_layernorm_bwd_ex(dtype, M, N, dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, scratch, nullptr,
                                 stream);
}

extern "C" int emoasr_layernorm_bwd_finalize(int n, const emoasr_ln_finalize_item_t* items, void* stream) {
  EMO_CHECK(n >= 0 && n <= EMOASR_LN_FINALIZE_MAX, "layernorm_bwd_finalize: n=%d outside 0..%d", n,
            EMOASR_LN_FINALIZE_MAX);
  if (n == 0) return 0;
  LnFinGroup G{};
  G.n = n;
  int maxn = 0, maxblk = 0;
  for (int i = 0; i < n; ++i) {
    EMO_CHECK(items[i].part && items[i].M > 0, "layernorm_bwd_finalize: bad item %d", i);
    G.it[i] = items[i];
    G.nblk[i] = ln_bwd_nblk(items[i].M, items[i].N);
    maxn = std::max(maxn, items[i].N);
    maxblk = std::max(maxblk, G.nblk[i]);
  }
  dim3 grid(cdiv(2 * maxn, 64), std::max(1, std::min(8, maxblk / 32)), n);
  ln_bwd_finalize_grouped_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(G);
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" long emoasr_layernorm_bwd_scratch_floats(int N) { return (long)LN_BWD8_MAXBLK * 2 * N; }
