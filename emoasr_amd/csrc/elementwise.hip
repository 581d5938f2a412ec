// Element-wise, layout and column-reduction helpers (all HBM-bound).
#include "common.h"
#include "../../include/emoasr_hip.h"

namespace {

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void strided_copy_kernel(const TI* __restrict__ in, TO* __restrict__ out,
                                                           int d1, int d2, int d3, long s0, long s1,
                                                           long s2, long s3, long n, int accumulate) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    long r = i;
    const int i3 = r % d3; r /= d3;
    const int i2 = r % d2; r /= d2;
    const int i1 = r % d1; r /= d1;
    const long src = r * s0 + i1 * s1 + i2 * s2 + i3 * s3;
    float v = to_f32(in[src]);
    if (accumulate) v += to_f32(out[i]);
    out[i] = from_f32<TO>(v);
  }
}

// dense source: 8 elements (one or two 16-byte accesses each way) per thread, no index arithmetic
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void dense_copy_kernel(const TI* __restrict__ in, TO* __restrict__ out, long n8,
                                                         int accumulate) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    float v[8];
    load8<TI>(in + i * 8, v);
    if (accumulate) {
      float o[8];
      load8<TO>(out + i * 8, o);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += o[e];
    }
    store8<TO>(out + i * 8, v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void scale_dropout_kernel(long n, const T* __restrict__ x,
                                                            T* __restrict__ y, float scale, float p,
                                                            uint64_t seed) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    y[i] = from_f32<T>(to_f32(x[i]) * scale * dropout_scale(seed, (uint64_t)i, p));
}

template <typename T>
__global__ __launch_bounds__(256) void posenc_kernel(int T_, int N, long n, const T* __restrict__ x,
                                                     const float* __restrict__ pe, float scale, float p,
                                                     uint64_t seed, T* __restrict__ y) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int col = i % N;
    const int t = (i / N) % T_;
    float v = to_f32(x[i]) * scale;
    if (pe) v += pe[(long)t * N + col];
    y[i] = from_f32<T>(v * dropout_scale(seed, (uint64_t)i, p));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void add_kernel(long n, const T* __restrict__ a, const T* __restrict__ b,
                                                  T* __restrict__ y) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    y[i] = from_f32<T>(to_f32(a[i]) + to_f32(b[i]));
}

// out[n] += scale * sum_m X[m, n]; block = 256 columns x ROWS rows, atomics across row blocks.
constexpr int CS_ROWS = 64;
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(int M, int N, const T* __restrict__ X, long ldx,
                                                     float* __restrict__ out, float scale) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= N) return;
  const int r0 = blockIdx.y * CS_ROWS, r1 = min(M, r0 + CS_ROWS);
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += to_f32(X[(long)r * ldx + col]);
  atomicAdd(&out[col], s * scale);
}

template <typename T>
__global__ __launch_bounds__(256) void glu_fwd_kernel(long n, int C, const T* __restrict__ in,
                                                      T* __restrict__ out) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long m = i / C; const int c = i % C;
    const float a = to_f32(in[m * 2 * C + c]), g = to_f32(in[m * 2 * C + C + c]);
    out[i] = from_f32<T>(a * sigmoid_t<T>(g));
  }
}
template <typename T>
__global__ __launch_bounds__(256) void glu_bwd_kernel(long n, int C, const T* __restrict__ in,
                                                      const T* __restrict__ dout, T* __restrict__ din) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long m = i / C; const int c = i % C;
    const float a = to_f32(in[m * 2 * C + c]), g = to_f32(in[m * 2 * C + C + c]);
    const float s = sigmoid_t<T>(g), d = to_f32(dout[i]);
    din[m * 2 * C + c] = from_f32<T>(d * s);
    din[m * 2 * C + C + c] = from_f32<T>(d * a * s * (1.f - s));
  }
}

inline int ew_grid(long n) { long b = (n + 255) / 256; return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

}  // namespace

extern "C" int emoasr_strided_copy(int dtype_in, int dtype_out, const void* in, void* out, int d0, int d1,
                                   int d2, int d3, long s0, long s1, long s2, long s3, int accumulate,
                                   void* stream) {
  const long n = (long)d0 * d1 * d2 * d3;
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  // a dense source (strides of a contiguous [d0,d1,d2,d3] array) with n % 8 == 0 and 16-byte aligned buffers
  // takes the vector path: the bf16 shadow of the whole parameter arena is refreshed through here every step
  const bool dense = (d3 == 1 || s3 == 1) && (d2 == 1 || s2 == d3) && (d1 == 1 || s1 == (long)d2 * d3) &&
                     (d0 == 1 || s0 == (long)d1 * d2 * d3) && n % 8 == 0 &&
                     (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
#define SC(TI, TO)                                                                                          \
  do {                                                                                                      \
    if (dense) dense_copy_kernel<TI, TO><<<ew_grid(n / 8), 256, 0, s>>>((const TI*)in, (TO*)out, n / 8, accumulate); \
    else strided_copy_kernel<TI, TO><<<ew_grid(n), 256, 0, s>>>((const TI*)in, (TO*)out, d1, d2, d3, s0, s1, s2, s3, n, accumulate); \
  } while (0)
  if (dtype_in == EMO_F32X3) dtype_in = EMO_F32;
  if (dtype_out == EMO_F32X3) dtype_out = EMO_F32;
  if (dtype_in == EMO_F32 && dtype_out == EMO_F32) SC(float, float);
  else if (dtype_in == EMO_F32 && dtype_out == EMO_BF16) SC(float, bf16);
  else if (dtype_in == EMO_BF16 && dtype_out == EMO_F32) SC(bf16, float);
  else if (dtype_in == EMO_BF16 && dtype_out == EMO_BF16) SC(bf16, bf16);
  else { emo_set_error("strided_copy: bad dtypes"); return 1; }
#undef SC
  EMO_LAUNCH_CHECK();
  return 0;
}

namespace {
struct TcGroup {
  int n;
  int tile0[EMOASR_TC_MAX + 1];   // first 32 x 32 tile of every item
  emoasr_tc_item_t it[EMOASR_TC_MAX];
};
// one 32 x 32 tile per block through LDS: coalesced f32 reads along the source rows, coalesced stores along the destination rows
template <typename TO>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const TcGroup G) {
  __shared__ float t[32][33];
  int i = 0;
  while (i + 1 < G.n && (int)blockIdx.x >= G.tile0[i + 1]) ++i;
  const emoasr_tc_item_t& q = G.it[i];
  const int tl = blockIdx.x - G.tile0[i], tc = (q.cols + 31) / 32;
  const int r0 = (tl / tc) * 32, c0 = (tl % tc) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    t[ty + 8 * k][tx] = (r < q.rows && c < q.cols) ? q.src[(long)r * q.cols + c] : 0.f;
  }
  __syncthreads();
  TO* dst = static_cast<TO*>(q.dst);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (c < q.cols && r < q.rows) dst[(long)c * q.ld_dst + r] = from_f32<TO>(t[tx][ty + 8 * k]);
  }
}
}  // namespace

extern "C" int emoasr_transpose_cast_batched(int dtype_out, int n, const emoasr_tc_item_t* items, void* stream) {
  EMO_CHECK(n >= 0 && n <= EMOASR_TC_MAX, "transpose_cast_batched: n=%d outside 0..%d", n, EMOASR_TC_MAX);
  if (n == 0) return 0;
  TcGroup G{};
  G.n = n;
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    EMO_CHECK(items[i].src && items[i].dst && items[i].rows > 0 && items[i].cols > 0 && items[i].ld_dst >= items[i].rows,
              "transpose_cast_batched: bad item %d", i);
    G.it[i] = items[i];
    G.tile0[i] = tiles;
    tiles += ((items[i].rows + 31) / 32) * ((items[i].cols + 31) / 32);
  }
  G.tile0[n] = tiles;
  if (dtype_out == EMO_BF16) transpose_cast_kernel<bf16><<<tiles, 256, 0, (hipStream_t)stream>>>(G);
  else if (emo_is_f32(dtype_out)) transpose_cast_kernel<float><<<tiles, 256, 0, (hipStream_t)stream>>>(G);
  else { emo_set_error("transpose_cast_batched: bad dtype"); return 1; }
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_scale_dropout(int dtype, long n, const void* x, void* y, float scale, float drop_p,
                                    uint64_t seed, void* stream) {
  if (n == 0) return 0;
  EMO_DISPATCH(dtype, (scale_dropout_kernel<T><<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(
                          n, (const T*)x, (T*)y, scale, drop_p, seed)));
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_posenc(int dtype, int B, int T_, int N, const void* x, const float* pe, float scale,
                             float drop_p, uint64_t seed, void* y, void* stream) {
  const long n = (long)B * T_ * N;
  if (n == 0) return 0;
  EMO_DISPATCH(dtype, (posenc_kernel<T><<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(
                          T_, N, n, (const T*)x, pe, scale, drop_p, seed, (T*)y)));
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_add(int dtype, long n, const void* a, const void* b, void* y, void* stream) {
  if (n == 0) return 0;
  EMO_DISPATCH(dtype, (add_kernel<T><<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(n, (const T*)a,
                                                                                 (const T*)b, (T*)y)));
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_colsum(int dtype, int M, int N, const void* X, long ldx, float* out, float scale,
                             int accumulate, void* stream) {
  if (!accumulate) hipMemsetAsync(out, 0, sizeof(float) * N, (hipStream_t)stream);
  if (M == 0) return 0;
  dim3 grid(cdiv(N, 256), cdiv(M, CS_ROWS));
  EMO_DISPATCH(dtype, (colsum_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>(M, N, (const T*)X, ldx,
                                                                              out, scale)));
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_glu_fwd(int dtype, int M, int C, const void* in, void* out, void* stream) {
  const long n = (long)M * C;
  if (n == 0) return 0;
  EMO_DISPATCH(dtype, (glu_fwd_kernel<T><<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(n, C, (const T*)in,
                                                                                     (T*)out)));
  EMO_LAUNCH_CHECK();
  return 0;
}
extern "C" int emoasr_glu_bwd(int dtype, int M, int C, const void* in, const void* dout, void* din,
                              void* stream) {
  const long n = (long)M * C;
  if (n == 0) return 0;
  EMO_DISPATCH(dtype, (glu_bwd_kernel<T><<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(
                          n, C, (const T*)in, (const T*)dout, (T*)din)));
  EMO_LAUNCH_CHECK();
  return 0;
}
