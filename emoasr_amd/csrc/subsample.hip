// Conv2d subsampling front-end, first layer and the col2im half of the second layer's
// data gradient (asr/modeling/encoders/conv.py:9-15,20-23).
//
// conv1: Conv2d(1 -> C, k3, s2) + ReLU on x[B,T,F] (f32 features).  One block per
// (b, t1): the three input rows it needs live in LDS, each thread owns one output
// channel (9 weights in registers) and writes channels-last y1[b,t1,f1,:] rows, so
// every store is a full coalesced C-wide segment.  conv2 runs as an implicit GEMM
// over this layout (gemm.hip).
#include "common.h"
#include "../../include/emoasr_hip.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void conv1_fwd_kernel(int Tn, int F, int T1, int F1, int C,
                                                        const float* __restrict__ x,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias,
                                                        T* __restrict__ y1) {
  extern __shared__ __attribute__((aligned(16))) float rows[];  // [3][F]
  const int b = blockIdx.x / T1, t1 = blockIdx.x % T1;
  const float* xb = x + ((long)b * Tn + 2 * t1) * F;
  for (int i = threadIdx.x; i < 3 * F; i += blockDim.x) rows[i] = xb[i];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float wr[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) wr[j] = w[c * 9 + j];
    const float bv = bias[c];
    T* yo = y1 + ((long)b * T1 + t1) * F1 * C + c;
    for (int f1 = 0; f1 < F1; ++f1) {
      float acc = bv;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc += wr[kh * 3 + kw] * rows[kh * F + 2 * f1 + kw];
      yo[(long)f1 * C] = from_f32<T>(fmaxf(acc, 0.f));
    }
  }
}

// The training shape (bf16, C = 256): one block per PAIR of output rows (five input rows in LDS), a thread owns TWO adjacent
// channels of one of the two rows -- the nine broadcast LDS reads of a window feed 18 multiply-adds instead of 9, and a wave's
// store is 256 contiguous bytes instead of 128 (round 4: 90 us per micro-batch, 3.1 TB/s of writes, with one channel per thread).
// Same arithmetic per output (bias, then the nine taps in kh, kw order: bit-identical to the kernel above).
__global__ __launch_bounds__(256) void conv1_fwd2_kernel(int Tn, int F, int T1, int F1, const float* __restrict__ x,
                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                         bf16* __restrict__ y1) {
  constexpr int C = 256;
  extern __shared__ __attribute__((aligned(16))) float rows[];  // [5][F]
  const int np = (T1 + 1) / 2;
  const int b = blockIdx.x / np, t1a = (blockIdx.x % np) * 2;
  const int nrow = min(5, Tn - 2 * t1a);   // (the second output row may not exist)
  const float* xb = x + ((long)b * Tn + 2 * t1a) * F;
  for (int i = threadIdx.x; i < nrow * F; i += 256) rows[i] = xb[i];
  __syncthreads();
  const int half = threadIdx.x >> 7, c = (threadIdx.x & 127) * 2;
  const int t1 = t1a + half;
  if (t1 >= T1) return;
  float w0[9], w1[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) { w0[j] = w[c * 9 + j]; w1[j] = w[(c + 1) * 9 + j]; }
  const float b0 = bias[c], b1 = bias[c + 1];
  const float* rw = rows + 2 * half * F;
  bf16* yo = y1 + ((long)b * T1 + t1) * F1 * C + c;
  for (int f1 = 0; f1 < F1; ++f1) {
    float a0 = b0, a1 = b1;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const float xv = rw[kh * F + 2 * f1 + kw];
        a0 += w0[kh * 3 + kw] * xv;
        a1 += w1[kh * 3 + kw] * xv;
      }
    bf16x2 o;
    o[0] = (bf16)fmaxf(a0, 0.f); o[1] = (bf16)fmaxf(a1, 0.f);
    *reinterpret_cast<bf16x2*>(yo + (long)f1 * C) = o;
  }
}

// dw1[c, kh*3+kw] += sum dy1[b,t1,f1,c] * x[b,2t1+kh,2f1+kw];  db1[c] += sum dy1
// Block = (b, chunk of C1_TROWS t1 rows, 256 channels); thread = 8 channels x one of 8 f1 lanes
// (the f1 lanes of a channel group are adjacent lanes: 16-byte dy1 loads, 128-byte row segments, and a
// 3-step shuffle folds them at the end).  The input rows of C1_SUB t1 rows are staged in LDS at a time.
// One atomic per (block, output): large row chunks keep the same-address contention low.
#ifndef EMO_C1_TROWS
#define EMO_C1_TROWS 32
#endif
constexpr int C1_TROWS = EMO_C1_TROWS;   // (-DEMO_C1_TROWS=16 | 8: build variants for the A/B of tools/r05_run24.sh)
constexpr int C1_SUB = 8;
constexpr int C1_FJ = 8;   // f1 positions per lane (8 lanes stride 8): F1 <= 64
template <typename T>
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(int Tn, int F, int T1, int F1, int C,
                                                          const float* __restrict__ x,
                                                          const T* __restrict__ dy1,
                                                          float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float rows[];  // [2*C1_SUB+1][F]
  const int nchunk = (T1 + C1_TROWS - 1) / C1_TROWS;
  const int b = blockIdx.x / nchunk, tc = blockIdx.x % nchunk;
  const int fl = threadIdx.x & 7, c = blockIdx.y * 256 + (threadIdx.x >> 3) * 8;
  const bool cok = c < C;
  float acc[9][8], sb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sb[e] = 0.f;
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j][e] = 0.f;
  }
  const __amdgpu_buffer_rsrc_t rsd = make_rsrc(dy1 + (long)b * T1 * F1 * C);
  const int t_end = min(T1, (tc + 1) * C1_TROWS);
  for (int ts = tc * C1_TROWS; ts < t_end; ts += C1_SUB) {
    const int nt = min(C1_SUB, t_end - ts);
    const float* xb = x + ((long)b * Tn + 2 * ts) * F;
    __syncthreads();
    for (int i = threadIdx.x; i < (2 * nt + 1) * F; i += 256) rows[i] = xb[i];
    __syncthreads();
    // all loads of one t1 row (up to C1_FJ f1 positions per lane) are issued before the first is used, and
    // the next row's are issued before this row is multiplied: with one wave per SIMD an un-pipelined loop
    // paid a full memory round trip per 16-byte load (~300 of them per thread)
    constexpr int NV = 8 * (int)sizeof(T) / 16;  // 16-byte pieces per 8 channels (1 for bf16, 2 for f32)
    Vec16<T> dn[C1_FJ][NV];
    auto fetch_row = [&](int tt) {
#pragma unroll
      for (int j = 0; j < C1_FJ; ++j) {
        const int f1 = fl + 8 * j;
        const bool ok = cok && tt < nt && f1 < F1;
        const long eo = ((long)(ts + tt) * F1 + f1) * C + c;
#pragma unroll
        for (int p = 0; p < NV; ++p)
          dn[j][p] = buf_load16<T>(rsd, ok ? (unsigned)((eo * sizeof(T)) + 16 * p) : EMO_OOB);
      }
    };
    fetch_row(0);
    for (int tt = 0; tt < nt; ++tt) {
      const float* r0 = rows + 2 * tt * F;
      Vec16<T> dc[C1_FJ][NV];
#pragma unroll
      for (int j = 0; j < C1_FJ; ++j)
#pragma unroll
        for (int p = 0; p < NV; ++p) dc[j][p] = dn[j][p];
      fetch_row(tt + 1);
#pragma unroll
      for (int j = 0; j < C1_FJ; ++j) {
        const int f1 = fl + 8 * j;
        if (f1 >= F1) break;
        float d[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] = dc[j][e / Vec16<T>::N].get(e % Vec16<T>::N);
        float xv[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) xv[kh * 3 + kw] = r0[kh * F + 2 * f1 + kw];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          sb[e] += d[e];
#pragma unroll
          for (int q = 0; q < 9; ++q) acc[q][e] += d[e] * xv[q];
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) sb[e] += __shfl_xor(sb[e], o, 64);
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) acc[j][e] += __shfl_xor(acc[j][e], o, 64);
  }
  // per-block partial sums [blk][C][10] (9 taps + bias): 260-way same-address float atomics cost more than
  // the whole product; conv1_wgrad_reduce_kernel folds the partials
  if (cok && fl == 0) {
    float* p = part + ((long)blockIdx.x * C + c) * 10;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int j = 0; j < 9; ++j) p[e * 10 + j] = acc[j][e];
      p[e * 10 + 9] = sb[e];
    }
  }
}

// dw[c][j] (+)= sum_blk part[blk][c][j] (j < 9), db[c] (+)= sum_blk part[blk][c][9]
__global__ __launch_bounds__(256) void conv1_wgrad_reduce_kernel(int nblk, int C, const float* __restrict__ part,
                                                                 float* __restrict__ dw, float* __restrict__ db,
                                                                 int accumulate) {
  __shared__ float red[4][64];
  const int li = threadIdx.x & 63, sl = threadIdx.x >> 6;  // 64 outputs x 4 slices of the partial list
  const int i = blockIdx.x * 64 + li;                      // (c, j) flattened
  float s = 0.f;
  if (i < C * 10) {
#pragma unroll 8
    for (int k = sl; k < nblk; k += 4) s += part[(long)k * C * 10 + i];
  }
  red[sl][li] = s;
  __syncthreads();
  if (sl != 0 || i >= C * 10) return;
  s = red[0][li] + red[1][li] + red[2][li] + red[3][li];
  const int c = i / 10, j = i - c * 10;
  float* o = j < 9 ? &dw[c * 9 + j] : &db[c];
  *o = accumulate ? *o + s : s;
}

// dy1[b,t1,f1,c] = [y1 > 0] * sum over (kh,kw) with t1 = 2*t2+kh, f1 = 2*f2+kw of
//                  dcol[(b,t2,f2), (kh*3+kw)*C + c]
// One thread = 8 consecutive channels of one (b,t1,f1) position (16-byte accesses).  All nine taps are
// loaded unconditionally with bounds-checked buffer loads: taps of the wrong parity or outside the
// output grid get an out-of-range offset (zeros, no traffic), so the loads are in flight together.
template <typename T>
__global__ __launch_bounds__(256) void col2im_kernel(int T1, int F1, int T2, int F2, int C, long nvec,
                                                     const T* __restrict__ dcol, const T* __restrict__ y1,
                                                     T* __restrict__ dy1) {
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(dcol);
  const int cv = C / 8;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long)gridDim.x * 256) {
    const int c = (int)(v % cv) * 8;
    long r = v / cv;
    const int f1 = r % F1; r /= F1;
    const int t1 = r % T1;
    const int b = r / T1;
    float yv[8], s[8];
    load8<T>(y1 + v * 8, yv);
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int tt = t1 - kh, t2 = tt >> 1;
      const bool tok = tt >= 0 && !(tt & 1) && t2 < T2;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ff = f1 - kw, f2 = ff >> 1;
        const bool ok = tok && ff >= 0 && !(ff & 1) && f2 < F2;
        float d[8];
        buf_load8<T>(rs, (((long)b * T2 + t2) * F2 + f2) * (9L * C) + (kh * 3 + kw) * C + c, ok, d);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += d[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = yv[e] > 0.f ? s[e] : 0.f;
    store8<T>(dy1 + v * 8, s);
  }
}

int g_conv1_pair = 1;   // option "conv1_pair"

inline int ew_grid(long n) { long b = (n + 255) / 256; return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b)); }

}  // namespace

void emo_conv1_set_pair(int v) { g_conv1_pair = v ? 1 : 0; }

extern "C" int emoasr_conv1_fwd(int dtype, int B, int Tn, int F, int C, const float* x, const float* w1,
                                const float* b1, void* y1, void* stream) {
  EMO_CHECK(Tn >= 3 && F >= 3, "conv1: input too small (T=%d F=%d)", Tn, F);
  const int T1 = (Tn - 3) / 2 + 1, F1 = (F - 3) / 2 + 1;
  if (B == 0) return 0;
  if (dtype == EMO_BF16 && C == 256 && g_conv1_pair) {
    conv1_fwd2_kernel<<<B * ((T1 + 1) / 2), 256, 5 * F * sizeof(float), (hipStream_t)stream>>>(Tn, F, T1, F1, x, w1, b1, (bf16*)y1);
    EMO_LAUNCH_CHECK();
    return 0;
  }
  EMO_DISPATCH(dtype, (conv1_fwd_kernel<T><<<B * T1, 256, 3 * F * sizeof(float), (hipStream_t)stream>>>(
                          Tn, F, T1, F1, C, x, w1, b1, (T*)y1)));
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" long emoasr_conv1_wgrad_scratch_floats(int B, int Tn, int C) {
  const int T1 = (Tn - 3) / 2 + 1;
  return (long)B * cdiv(T1 > 0 ? T1 : 1, C1_TROWS) * C * 10;
}

extern "C" int emoasr_conv1_wgrad(int dtype, int B, int Tn, int F, int C, const float* x, const void* dy1,
                                  float* dw1, float* db1, int accumulate, float* scratch, void* stream) {
  EMO_CHECK(Tn >= 3 && F >= 3, "conv1_wgrad: input too small");
  EMO_CHECK(C <= 1024, "conv1_wgrad: C=%d > 1024", C);
  const int T1 = (Tn - 3) / 2 + 1, F1 = (F - 3) / 2 + 1;
  hipStream_t s = (hipStream_t)stream;
  if (B == 0) {
    if (!accumulate) {
      hipMemsetAsync(dw1, 0, sizeof(float) * C * 9, s);
      hipMemsetAsync(db1, 0, sizeof(float) * C, s);
    }
    return 0;
  }
  EMO_CHECK(C % 8 == 0, "conv1_wgrad: C must be a multiple of 8");
  EMO_CHECK(F1 <= 8 * C1_FJ, "conv1_wgrad: F1=%d > %d", F1, 8 * C1_FJ);
  EMO_CHECK(scratch, "conv1_wgrad: scratch of emoasr_conv1_wgrad_scratch_floats(B, T, C) floats required");
  const int nchunk = cdiv(T1, C1_TROWS);
  dim3 grid(B * nchunk, cdiv(C, 256));
  EMO_DISPATCH(dtype, (conv1_wgrad_kernel<T><<<grid, 256, (2 * C1_SUB + 1) * F * sizeof(float), s>>>(
                          Tn, F, T1, F1, C, x, (const T*)dy1, scratch)));
  conv1_wgrad_reduce_kernel<<<cdiv(C * 10, 64), 256, 0, s>>>(B * nchunk, C, scratch, dw1, db1, accumulate);
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_conv2_col2im(int dtype, int B, int T1, int F1, int C, const void* dcol,
                                   const void* y1, void* dy1, void* stream) {
  EMO_CHECK(T1 >= 3 && F1 >= 3, "col2im: input too small");
  const int T2 = (T1 - 3) / 2 + 1, F2 = (F1 - 3) / 2 + 1;
  const long n = (long)B * T1 * F1 * C;
  if (n == 0) return 0;
  EMO_CHECK(C % 8 == 0, "col2im: C must be a multiple of 8");
  EMO_CHECK((long)B * T2 * F2 * 9 * C * (dtype == EMO_BF16 ? 2 : 4) < (1L << 32), "col2im: dcol larger than 4 GiB");
  EMO_DISPATCH(dtype, (col2im_kernel<T><<<ew_grid(n / 8), 256, 0, (hipStream_t)stream>>>(
                          T1, F1, T2, F2, C, n / 8, (const T*)dcol, (const T*)y1, (T*)dy1)));
  EMO_LAUNCH_CHECK();
  return 0;
}
