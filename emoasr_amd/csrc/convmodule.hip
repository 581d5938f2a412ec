// Conformer convolution module pieces (asr/modeling/conformer.py:98-143):
// depthwise Conv1d(k=31, groups=C) over time, BatchNorm1d (training: batch statistics
// over ALL B*T rows -- the reference never masks padded frames), Swish.
// Channels-last activations [B, T, C]: threads run along C so every global access
// is a coalesced row segment; the time window slides in registers.
#include <algorithm>
#include "common.h"
#include "../../include/emoasr_hip.h"

namespace {

constexpr int DW_TT = 32;    // output frames per block (16: twice the blocks, measured slower -- more halo loads and partials)
constexpr int DW_MAXK = 31;  // kernel taps (compile-time bound of the register window)
constexpr int DW_WCHUNKS = 1; // time chunks per block in the weight-gradient kernel

// y[b,t,c] = bias[c] + sum_j w[c,j] * x[b, t + j - pad, c]      (flip=0)
// dx[b,t,c] =          sum_j w[c,K-1-j] * dy[b, t + j - pad, c]  (flip=1, no bias)
// The time window is loaded with bounds-checked buffer loads (common.h): all DW_TT + K - 1 guarded
// loads of a thread are in flight together.  `part` (optional, forward only): per-block BatchNorm
// partial statistics [block][2][C] = (sum, centred sum of squares about the block mean) of the
// STORED (rounded to T) outputs of this block's rows; emoasr_bn_stats_finalize merges them.
template <typename T>
__global__ __launch_bounds__(256) void dwconv_kernel(int Tn, int C, int K, const T* __restrict__ x,
                                                     const float* __restrict__ w,
                                                     const float* __restrict__ bias, T* __restrict__ y,
                                                     int flip, float* __restrict__ part, const RowSegs sg) {
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= C) return;
  int b = blockIdx.z;
  if (sg.n > 1) {   // stacked micro-batches: this utterance's segment (own padded length, own area of the partial table)
    const int si = rowsegs_of_utt(sg, b);
    Tn = sg.T[si];
    x += sg.row[si] * C; y += sg.row[si] * C;
    if (part) part += sg.part[si];
    b -= sg.b0[si];
    if ((int)blockIdx.x * DW_TT >= Tn) return;   // (the grid follows the longest segment)
  }
  const int nx = (Tn + DW_TT - 1) / DW_TT;
  const int t0 = blockIdx.x * DW_TT, pad = (K - 1) / 2;
  float wr[DW_MAXK];
#pragma unroll
  for (int j = 0; j < DW_MAXK; ++j) wr[j] = j < K ? w[c * K + (flip ? K - 1 - j : j)] : 0.f;
  const float bv = bias ? bias[c] : 0.f;
  float win[DW_TT + DW_MAXK - 1];
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(x + (long)b * Tn * C);  // one utterance: < 4 GiB
#pragma unroll
  for (int i = 0; i < DW_TT + DW_MAXK - 1; ++i) {
    const int t = t0 + i - pad;
    const bool ok = i < DW_TT + K - 1 && t >= 0 && t < Tn;
    win[i] = buf_load_f32<T>(rs, ok ? (unsigned)(((long)t * C + c) * sizeof(T)) : EMO_OOB);
  }
  T* yb = y + (long)b * Tn * C + c;
  float out[DW_TT];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < DW_TT; ++i) {
    float acc = bv;
#pragma unroll
    for (int j = 0; j < DW_MAXK; ++j) acc = emo_mac(wr[j], win[i + j], acc);
    const T r = from_f32<T>(acc);
    out[i] = t0 + i < Tn ? to_f32(r) : 0.f;
    s += out[i];
    if (t0 + i < Tn) yb[(long)(t0 + i) * C] = r;
  }
  if (part) {
    const int n = min(DW_TT, Tn - t0);
    const float mb = s / n;
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < DW_TT; ++i) {
      const float d = out[i] - mb;
      m2 += i < n ? d * d : 0.f;
    }
    float* p = part + ((long)b * nx + blockIdx.x) * 2 * C + c;
    p[0] = s;
    p[C] = m2;
  }
}

// Merge of the per-block partials (Chan et al.): mean = sum_b s_b / M,
// M2 = sum_b [ m2_b + n_b (s_b / n_b - mean)^2 ]; var = M2 / M (biased, used to normalise),
// running_var gets the unbiased M2 / (M - 1) like nn.BatchNorm1d.  One block = 16 channels x 64
// groups of partial blocks (16 blocks for C = 256: the partial table is walked in 7 steps, not 28).
constexpr int BNF_CH = 16, BNF_G = 64;  // channels x groups of partial blocks per finalize block
// arrival tickets per channel group ([0][64] statistics, [1][64] backward fold; zero between launches): `tickets` argument, one
// area per (device, stream) from emo_stream_scratch -- a __device__ array was shared by every stream of the process
#define EMO_BN_TICKETS(stream_)                                                                   \
  EmoScratch* tsc_ = emo_stream_scratch(EMO_SCRATCH_BN_TICKETS, (void*)(stream_), 2 * 64 * sizeof(unsigned)); \
  if (!tsc_) return 1;                                                                            \
  unsigned* tickets_ = static_cast<unsigned*>(tsc_->dev)
__global__ __launch_bounds__(1024) void bn_stats_finalize_kernel(int B_, int Tn_, int C, const float* __restrict__ part_,
                                                                 float* __restrict__ mean_, float* __restrict__ var_,
                                                                 float* __restrict__ running_mean,
                                                                 float* __restrict__ running_var, float momentum,
                                                                 long long* __restrict__ num_batches_tracked, const RowSegs sg,
                                                                 unsigned* __restrict__ tickets) {
  __shared__ float red[BNF_G][BNF_CH];
  __shared__ float mean_s[BNF_CH];
  __shared__ int s_last;
  const int lane = threadIdx.x % BNF_CH, grp = threadIdx.x / BNF_CH;
  const int c = blockIdx.x * BNF_CH + lane;
  // stacked micro-batches: blockIdx.y = segment (one set of statistics each, computed in parallel); the running statistics must
  // move once per segment IN ORDER (the arithmetic of the separate passes): the block of a channel group that arrives last
  // applies all segments' updates.  sg.n <= 1: the one dense batch (B_, Tn_).
  const int ns = sg.n > 1 ? sg.n : 1, si = blockIdx.y;
  const int B = sg.n > 1 ? sg.b0[si + 1] - sg.b0[si] : B_, Tn = sg.n > 1 ? sg.T[si] : Tn_;
  const float* part = sg.n > 1 ? part_ + sg.part[si] : part_;
  float* mean = mean_ + (long)si * C;
  float* var = var_ + (long)si * C;
  const int nx = (Tn + DW_TT - 1) / DW_TT, nblk = B * nx;
  const float M = (float)B * Tn;
  float s = 0.f;
  if (c < C) {
#pragma unroll 8
    for (int k = grp; k < nblk; k += BNF_G) s += part[(long)k * 2 * C + c];
  }
  red[grp][lane] = s;
  __syncthreads();
  if (grp == 0) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < BNF_G; ++g) t += red[g][lane];
    mean_s[lane] = t / M;
  }
  __syncthreads();
  const float mu = mean_s[lane];
  float m2 = 0.f;
  if (c < C)
#pragma unroll 8
    for (int k = grp; k < nblk; k += BNF_G) {
      const int n = min(DW_TT, Tn - (k % nx) * DW_TT);
      const float d = part[(long)k * 2 * C + c] / n - mu;
      m2 += part[(long)k * 2 * C + C + c] + n * d * d;
    }
  __syncthreads();
  red[grp][lane] = m2;
  __syncthreads();
  float m2sum = 0.f;   // this segment's centred sum of squares (lanes of group 0)
  if (grp == 0 && c < C) {
#pragma unroll
    for (int g = 0; g < BNF_G; ++g) m2sum += red[g][lane];
    mean[c] = mu;
    var[c] = m2sum / M;
  }
  if (!running_mean && !running_var && !num_batches_tracked) return;
  // ---- running statistics, in segment order, by the last block of this channel group ------------------------------------------------
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    s_last = atomicAdd(&tickets[blockIdx.x], 1u) == (unsigned)ns - 1;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  if (grp == 0 && c < C) {
    float rm = running_mean ? running_mean[c] : 0.f, rv = running_var ? running_var[c] : 0.f;
    for (int k = 0; k < ns; ++k) {
      const float Mk = sg.n > 1 ? (float)(sg.row[k + 1] - sg.row[k]) : M;
      const float mk = __hip_atomic_load(mean_ + (long)k * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float vk = __hip_atomic_load(var_ + (long)k * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // biased: M2 / M
      // (the own segment's sum is at hand exactly -- a dense batch updates bit for bit as before; the others' come back from var)
      const float m2k = k == si ? m2sum : vk * Mk;
      rm = (1.f - momentum) * rm + momentum * (k == si ? mu : mk);
      rv = (1.f - momentum) * rv + momentum * (Mk > 1.f ? m2k / (Mk - 1.f) : m2k / Mk);
    }
    if (running_mean) running_mean[c] = rm;
    if (running_var) running_var[c] = rv;
  }
  if (threadIdx.x == 0) {
    tickets[blockIdx.x] = 0u;
    if (num_batches_tracked && blockIdx.x == 0) *num_batches_tracked += ns;
  }
}

// dw[c,j] += sum_{b,t} dy[b,t,c] * x[b,t+j-pad,c];  dbias[c] += sum dy
template <typename T>
__global__ __launch_bounds__(256) void dwconv_bwd_w_kernel(int Tn, int C, int K, const T* __restrict__ dy,
                                                           const T* __restrict__ x, float* __restrict__ part, const RowSegs sg) {
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= C) return;
  int b = blockIdx.z;
  const int pad = (K - 1) / 2;
  if (sg.n > 1) {   // stacked micro-batches: this utterance's segment; a block past its segment's end leaves a zero partial row
    const int si = rowsegs_of_utt(sg, b);
    Tn = sg.T[si];
    x += sg.row[si] * C; dy += sg.row[si] * C;
    b -= sg.b0[si];
  }
  float acc[DW_MAXK];
#pragma unroll
  for (int j = 0; j < DW_MAXK; ++j) acc[j] = 0.f;
  float sb = 0.f;
  const __amdgpu_buffer_rsrc_t rsx = make_rsrc(x + (long)b * Tn * C), rsd = make_rsrc(dy + (long)b * Tn * C);
  // several time chunks per block: fewer same-address atomics (f32 atomics collapse ~14x when
  // every workgroup hits the same few KB)
  for (int cc = 0; cc < DW_WCHUNKS; ++cc) {
    const int t0 = (blockIdx.x * DW_WCHUNKS + cc) * DW_TT;
    if (t0 >= Tn) break;
    float win[DW_TT + DW_MAXK - 1];
#pragma unroll
    for (int i = 0; i < DW_TT + DW_MAXK - 1; ++i) {
      const int t = t0 + i - pad;
      const bool ok = i < DW_TT + K - 1 && t >= 0 && t < Tn;
      win[i] = buf_load_f32<T>(rsx, ok ? (unsigned)(((long)t * C + c) * sizeof(T)) : EMO_OOB);
    }
    float dv[DW_TT];
#pragma unroll
    for (int i = 0; i < DW_TT; ++i)
      dv[i] = buf_load_f32<T>(rsd, t0 + i < Tn ? (unsigned)(((long)(t0 + i) * C + c) * sizeof(T)) : EMO_OOB);
#pragma unroll
    for (int i = 0; i < DW_TT; ++i) {
      const float d = dv[i];
      sb += d;
#pragma unroll
      for (int j = 0; j < DW_MAXK; ++j) acc[j] = emo_mac(d, win[i + j], acc[j]);
    }
  }
  // per-block partials [blk][K+1][C] (coalesced along c); folded by dwconv_bwd_w_reduce_kernel
  const long blk = ((long)blockIdx.z * gridDim.x + blockIdx.x);
  float* p = part + blk * (K + 1) * C + c;
#pragma unroll
  for (int j = 0; j < DW_MAXK; ++j)
    if (j < K) p[(long)j * C] = acc[j];
  p[(long)K * C] = sb;
}

__global__ __launch_bounds__(256) void dwconv_bwd_w_reduce_kernel(int nblk, int C, int K,
                                                                  const float* __restrict__ part,
                                                                  float* __restrict__ dw,
                                                                  float* __restrict__ dbias) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;  // over (K+1)*C, laid out [j][c]
  const int n = (K + 1) * C;
  // blockIdx.y = a quarter (gridDim.y-th) of the partial rows: 128 workgroups could not pull the ~80 MB of partials of a stacked
  // launch faster than 2.4 TB/s; the parts meet in f32 atomics (the weight gradients of the GEMMs arrive the same way)
  const int per = (nblk + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
  float s = 0.f;
  if (i < n) {
#pragma unroll 8
    for (int b = b0 + wave; b < b1; b += 4) s += part[(long)b * n + i];
  }
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && i < n) {
    s = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    const int j = i / C, c = i % C;
    if (gridDim.y == 1) {
      if (j < K) dw[c * K + j] += s;
      else if (dbias) dbias[c] += s;
    } else {
      if (j < K) atomicAdd(&dw[c * K + j], s);
      else if (dbias) atomicAdd(&dbias[c], s);
    }
  }
}

// ---- BatchNorm statistics: two passes (sum, then centred sum of squares) --------
constexpr int BN_ROWS = 16;
constexpr int BN_STAT_ROWS = 64;  // rows per block in the reduction kernels (contended f32 atomics)
template <typename T>
__global__ __launch_bounds__(256) void bn_sum_kernel(int M, int C, const T* __restrict__ y,
                                                     float* __restrict__ sum) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const int r0 = blockIdx.y * BN_STAT_ROWS, r1 = min(M, r0 + BN_STAT_ROWS);
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += to_f32(y[(long)r * C + c]);
  atomicAdd(&sum[c], s);
}
template <typename T>
__global__ __launch_bounds__(256) void bn_var_kernel(int M, int C, const T* __restrict__ y,
                                                     const float* __restrict__ sum,
                                                     float* __restrict__ sq) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float mu = sum[c] / M;
  const int r0 = blockIdx.y * BN_STAT_ROWS, r1 = min(M, r0 + BN_STAT_ROWS);
  float s = 0.f;
  for (int r = r0; r < r1; ++r) { const float d = to_f32(y[(long)r * C + c]) - mu; s += d * d; }
  atomicAdd(&sq[c], s);
}
// mean/var hold sum / centred-sq-sum on entry; finalise and update running stats.
__global__ void bn_finalize_kernel(int M, int C, float* mean, float* var, float* running_mean,
                                   float* running_var, float momentum) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float mu = mean[c] / M, vb = var[c] / M;
  mean[c] = mu; var[c] = vb;
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
  if (running_var) {
    const float vu = M > 1 ? var[c] * M / (M - 1) : vb;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * vu;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_swish_fwd_kernel(long n, int C, const T* __restrict__ y,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ var,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps,
                                                           T* __restrict__ z, const RowSegs sg) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = i % C;
    // stacked micro-batches: the statistics of the row's segment ([n, C] tables)
    const long so = sg.n > 1 ? (long)rowsegs_of_row(sg, i / C) * C : 0;
    const float xh = (to_f32(y[i]) - mean[so + c]) * rsqrtf(var[so + c] + eps);
    z[i] = from_f32<T>(swish_t<T>(gamma[c] * xh + beta[c]));
  }
}
// C % 8 == 0: eight channels of one row per thread and pass, 16-byte accesses (the element-wise form above ran at 1.4 TB/s);
// the same arithmetic per element, so the two agree bit for bit
template <typename T>
__global__ __launch_bounds__(256) void bn_swish_fwd8_kernel(long n8, int C, const T* __restrict__ y,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ var,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            T* __restrict__ z, const RowSegs sg) {
  const int cg = C / 8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const long row = i / cg;
    const int c = (int)(i - row * cg) * 8;
    const long so = sg.n > 1 ? (long)rowsegs_of_row(sg, row) * C : 0;
    float v[8];
    load8<T>(y + i * 8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = (v[j] - mean[so + c + j]) * rsqrtf(var[so + c + j] + eps);
      v[j] = swish_t<T>(gamma[c + j] * xh + beta[c + j]);
    }
    store8<T>(z + i * 8, v);
  }
}

// ---- BatchNorm + Swish backward (training statistics), C % 8 == 0 ------------------------------
// Layout of both passes: a block is 8 row-lanes x 32 channel groups of 8 channels (16-byte loads);
// rows r0 + lane_row + 8*it.  Pass 1 writes per-block partial sums [block][2][C] (no atomics, no
// zeroing); pass 2 first folds the few partial rows for its channels (block-cooperatively through
// LDS), then applies.
constexpr int BN_SUM_ROWS = 64;  // rows per block of pass 1 (-> M/64 partial rows, folded by bn_bwd_fold_kernel; 16 rows per block spent most of a block on its per-channel set-up: family 3.63 -> 3.42 ms per step)
constexpr int BN_APPLY_ROWS = 16; // rows per block of pass 3

// pass 1: part[blk][c] = sum dbn, part[blk][C+c] = sum dbn*xhat   with dbn = dz * swish'(bn)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_sums_kernel(int M, int C, const T* __restrict__ dz,
                                                          const T* __restrict__ y,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ var,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps,
                                                          float* __restrict__ part, const RowSegs sg) {
  __shared__ float red[8][2][256];
  const int hl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 256 + hl * 8;
  const bool cok = c < C;
  long rbase = 0;
  if (sg.n > 1) {   // stacked micro-batches: blockIdx.z = segment (its rows, its statistics, its area of the partial table)
    const int si = blockIdx.z;
    rbase = sg.row[si];
    M = (int)(sg.row[si + 1] - rbase);
    if ((int)blockIdx.y * BN_SUM_ROWS >= M) return;
    mean += (long)si * C; var += (long)si * C;
    part += sg.sums[si] * 2 * C;
  }
  float mu[8], is[8], g[8], bt[8], s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    mu[j] = cok ? mean[c + j] : 0.f; is[j] = cok ? rsqrtf(var[c + j] + eps) : 0.f;
    g[j] = cok ? gamma[c + j] : 0.f; bt[j] = cok ? beta[c + j] : 0.f;
    s1[j] = 0.f; s2[j] = 0.f;
  }
  const __amdgpu_buffer_rsrc_t rsy = make_rsrc(y), rsd = make_rsrc(dz);
  const int r0 = blockIdx.y * BN_SUM_ROWS;
#pragma unroll 4
  for (int it = 0; it < BN_SUM_ROWS / 8; ++it) {
    const int r = r0 + rl + 8 * it;
    const bool ok = cok && r < M;
    float yv[8], dv[8];
    buf_load8<T>(rsy, (rbase + r) * C + c, ok, yv);
    buf_load8<T>(rsd, (rbase + r) * C + c, ok, dv);  // dz = 0 past the end: contributes nothing
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = (yv[j] - mu[j]) * is[j];
      const float dbn = dv[j] * dswish_t<T>(g[j] * xh + bt[j]);
      s1[j] += dbn; s2[j] += dbn * xh;
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[rl][0][hl * 8 + j] = s1[j]; red[rl][1][hl * 8 + j] = s2[j]; }
  __syncthreads();
  const int cc = blockIdx.x * 256 + threadIdx.x;
  if (cc < C) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { a += red[k][0][threadIdx.x]; b += red[k][1][threadIdx.x]; }
    part[(long)blockIdx.y * 2 * C + cc] = a;
    part[(long)blockIdx.y * 2 * C + C + cc] = b;
  }
}
// pass 2: tot[c] = mean(dbn), tot[C + c] = mean(dbn * xhat) from the partial rows (16 channels x 64 groups of
// partial rows per block, like bn_stats_finalize_kernel); also dbeta += sum dbn, dgamma += sum dbn*xhat.
__global__ __launch_bounds__(1024) void bn_bwd_fold_kernel(int npart_, int C, float inv_m_, const float* __restrict__ part_,
                                                           float* __restrict__ tot_, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, const RowSegs sg,
                                                           unsigned* __restrict__ tickets) {
  __shared__ float red[2][64][16];
  __shared__ int s_last;
  const int lane = threadIdx.x % 16, grp = threadIdx.x / 16;
  const int c = blockIdx.x * 16 + lane;
  // stacked micro-batches: blockIdx.y = segment (one pair of means each: tot [n, 2, C], then the raw sums [n, 2, C]); dgamma / dbeta
  // take the segments' sums IN ORDER, added by the block of a channel group that arrives last (no float atomics)
  const int ns = sg.n > 1 ? sg.n : 1, si = blockIdx.y;
  const int npart = sg.n > 1 ? (int)(sg.sums[si + 1] - sg.sums[si]) : npart_;
  const float inv_m = sg.n > 1 ? 1.f / (float)(sg.row[si + 1] - sg.row[si]) : inv_m_;
  const float* part = sg.n > 1 ? part_ + sg.sums[si] * 2 * C : part_;
  float* tot = tot_ + (long)si * 2 * C;
  float* raw = tot_ + (long)ns * 2 * C + (long)si * 2 * C;   // (stacked launches only)
  float a = 0.f, b = 0.f;
  if (c < C) {
#pragma unroll 8
    for (int k = grp; k < npart; k += 64) { a += part[(long)k * 2 * C + c]; b += part[(long)k * 2 * C + C + c]; }
  }
  red[0][grp][lane] = a;
  red[1][grp][lane] = b;
  __syncthreads();
  if (grp == 0 && c < C) {
    a = 0.f; b = 0.f;
#pragma unroll
    for (int g = 0; g < 64; ++g) { a += red[0][g][lane]; b += red[1][g][lane]; }
    tot[c] = a * inv_m;
    tot[C + c] = b * inv_m;
    if (ns == 1) {
      if (dbeta) dbeta[c] += a;
      if (dgamma) dgamma[c] += b;
    } else {
      raw[c] = a;
      raw[C + c] = b;
    }
  }
  if (ns == 1 || (!dbeta && !dgamma)) return;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    s_last = atomicAdd(&tickets[64 + blockIdx.x], 1u) == (unsigned)ns - 1;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  if (grp == 0 && c < C) {
    float da = 0.f, db = 0.f;
    const float* r0 = tot_ + (long)ns * 2 * C;
    for (int k = 0; k < ns; ++k) {
      da += __hip_atomic_load(r0 + (long)k * 2 * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      db += __hip_atomic_load(r0 + (long)k * 2 * C + C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (dbeta) dbeta[c] += da;
    if (dgamma) dgamma[c] += db;
  }
  if (threadIdx.x == 0) tickets[64 + blockIdx.x] = 0u;
}
// pass 3: dy = gamma*invstd*(dbn - mean(dbn) - xhat*mean(dbn*xhat))
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(int M, int C, const T* __restrict__ dz,
                                                           const T* __restrict__ y,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ var,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps,
                                                           const float* __restrict__ tot,
                                                           T* __restrict__ dy, const RowSegs sg) {
  const int hl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 256 + hl * 8;
  if (c >= C) return;
  if (sg.n > 1) {   // stacked micro-batches: blockIdx.z = segment (own rows, statistics and means)
    const int si = blockIdx.z;
    M = (int)(sg.row[si + 1] - sg.row[si]);
    if ((int)blockIdx.y * BN_APPLY_ROWS >= M) return;
    dz += sg.row[si] * C; y += sg.row[si] * C; dy += sg.row[si] * C;
    mean += (long)si * C; var += (long)si * C; tot += (long)si * 2 * C;
  }
  float mu[8], is[8], g[8], bt[8], m1[8], m2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    mu[j] = mean[c + j]; is[j] = rsqrtf(var[c + j] + eps); g[j] = gamma[c + j]; bt[j] = beta[c + j];
    m1[j] = tot[c + j]; m2[j] = tot[C + c + j];
  }
  const __amdgpu_buffer_rsrc_t rsy = make_rsrc(y), rsd = make_rsrc(dz);
  const int r0 = blockIdx.y * BN_APPLY_ROWS;
#pragma unroll
  for (int it = 0; it < BN_APPLY_ROWS / 8; ++it) {
    const int r = r0 + rl + 8 * it;
    const bool ok = r < M;
    float yv[8], dv[8], o[8];
    buf_load8<T>(rsy, (long)r * C + c, ok, yv);
    buf_load8<T>(rsd, (long)r * C + c, ok, dv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = (yv[j] - mu[j]) * is[j];
      const float dbn = dv[j] * dswish_t<T>(g[j] * xh + bt[j]);
      o[j] = g[j] * is[j] * (dbn - m1[j] - xh * m2[j]);
    }
    if (ok) store8<T>(dy + (long)r * C + c, o);
  }
}

inline int ew_grid(long n) { long b = (n + 255) / 256; return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

}  // namespace

// convfused.hip: the same stencil with the time tile staged through LDS by 16-byte loads (bf16; bit-identical results)
int emo_dwconv_lds(int B, int Tn, int C, int K, const void* x, const float* w, const float* bias, void* y, int flip,
                   float* part, hipStream_t s);
static int g_dwconv_lds = 1;
void emo_conv_set_dwconv_lds(int v) { g_dwconv_lds = v; }
static bool use_lds(int dtype, int Tn, int C) {
  return g_dwconv_lds && dtype == EMO_BF16 && C % 8 == 0 && (long)Tn * C * 2 < (1L << 32);
}
int emo_dwconv_bwd_w_reduce(int nblk, int C, int K, const float* part, float* dw, float* dbias, hipStream_t s) {
  dwconv_bwd_w_reduce_kernel<<<dim3(cdiv((K + 1) * C, 64), nblk >= 1024 ? 4 : 1), 256, 0, s>>>(nblk, C, K, part, dw, dbias);
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_dwconv_fwd(int dtype, int B, int Tn, int C, int K, const void* x, const float* w,
                                 const float* bias, void* y, void* stream) {
  EMO_CHECK(K <= DW_MAXK && (K & 1), "dwconv: K=%d unsupported (odd, <= %d)", K, DW_MAXK);
  if (B * Tn == 0) return 0;
  if (use_lds(dtype, Tn, C)) return emo_dwconv_lds(B, Tn, C, K, x, w, bias, y, 0, nullptr, (hipStream_t)stream);
  dim3 grid(cdiv(Tn, DW_TT), cdiv(C, 256), B);
  EMO_DISPATCH(dtype, (dwconv_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>(Tn, C, K, (const T*)x, w,
                                                                              bias, (T*)y, 0, nullptr, RowSegs{})));
  EMO_LAUNCH_CHECK();
  return 0;
}

// Forward fused with the BatchNorm batch-statistics partials of its output (conformer.py:129-131:
// depthwise_conv -> batch_norm): `part` holds emoasr_dwconv_stats_floats() floats and is consumed by
// emoasr_bn_stats_finalize.  Replaces two memsets + two reduction passes over y.
extern "C" long emoasr_dwconv_stats_floats(int B, int Tn, int C) { return (long)B * cdiv(Tn, DW_TT) * 2 * C; }

extern "C" int emoasr_dwconv_fwd_stats(int dtype, int B, int Tn, int C, int K, const void* x, const float* w,
                                       const float* bias, void* y, float* part, void* stream) {
  EMO_CHECK(K <= DW_MAXK && (K & 1), "dwconv: K=%d unsupported", K);
  EMO_CHECK(part != nullptr && B * Tn > 0, "dwconv_fwd_stats: needs a non-empty batch and the partials buffer");
  if (use_lds(dtype, Tn, C)) return emo_dwconv_lds(B, Tn, C, K, x, w, bias, y, 0, part, (hipStream_t)stream);
  dim3 grid(cdiv(Tn, DW_TT), cdiv(C, 256), B);
  EMO_DISPATCH(dtype, (dwconv_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>(Tn, C, K, (const T*)x, w,
                                                                              bias, (T*)y, 0, part, RowSegs{})));
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_bn_stats_finalize(int B, int Tn, int C, const float* part, float* mean, float* var,
                                        float* running_mean, float* running_var, float momentum,
                                        long long* num_batches_tracked, void* stream) {
  EMO_CHECK(B * Tn > 0, "bn_stats_finalize: empty batch");
  EMO_CHECK(cdiv(C, BNF_CH) <= 64, "bn_stats_finalize: C=%d too wide for the ticket table", C);
  EMO_BN_TICKETS(stream);
  bn_stats_finalize_kernel<<<cdiv(C, BNF_CH), 1024, 0, (hipStream_t)stream>>>(B, Tn, C, part, mean, var, running_mean,
                                                                          running_var, momentum, num_batches_tracked, RowSegs{}, tickets_);
  EMO_LAUNCH_CHECK();
  return 0;
}
extern "C" int emoasr_dwconv_bwd_x(int dtype, int B, int Tn, int C, int K, const void* dy, const float* w,
                                   void* dx, void* stream) {
  EMO_CHECK(K <= DW_MAXK && (K & 1), "dwconv: K=%d unsupported", K);
  if (B * Tn == 0) return 0;
  if (use_lds(dtype, Tn, C)) return emo_dwconv_lds(B, Tn, C, K, dy, w, nullptr, dx, 1, nullptr, (hipStream_t)stream);
  dim3 grid(cdiv(Tn, DW_TT), cdiv(C, 256), B);
  EMO_DISPATCH(dtype, (dwconv_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>(Tn, C, K, (const T*)dy, w,
                                                                              nullptr, (T*)dx, 1, nullptr, RowSegs{})));
  EMO_LAUNCH_CHECK();
  return 0;
}
extern "C" long emoasr_dwconv_bwd_w_scratch_floats(int B, int Tn, int C, int K) {
  return (long)B * cdiv(Tn, DW_TT * DW_WCHUNKS) * (K + 1) * C;
}

extern "C" int emoasr_dwconv_bwd_w(int dtype, int B, int Tn, int C, int K, const void* dy, const void* x,
                                   float* dw, float* dbias, int accumulate, float* scratch, void* stream) {
  EMO_CHECK(K <= DW_MAXK && (K & 1), "dwconv: K=%d unsupported", K);
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate) {
    hipMemsetAsync(dw, 0, sizeof(float) * C * K, s);
    if (dbias) hipMemsetAsync(dbias, 0, sizeof(float) * C, s);
  }
  if (B * Tn == 0) return 0;
  EMO_CHECK(scratch != nullptr, "dwconv_bwd_w: scratch of emoasr_dwconv_bwd_w_scratch_floats() floats required");
  dim3 grid(cdiv(Tn, DW_TT * DW_WCHUNKS), cdiv(C, 256), B);
  EMO_DISPATCH(dtype, (dwconv_bwd_w_kernel<T><<<grid, 256, 0, s>>>(Tn, C, K, (const T*)dy, (const T*)x, scratch, RowSegs{})));
  const int nblk = grid.x * B;
  dwconv_bwd_w_reduce_kernel<<<cdiv((K + 1) * C, 64), 256, 0, s>>>(nblk, C, K, scratch, dw, dbias);
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_bn_stats(int dtype, int M, int C, const void* y, float* mean, float* var,
                               float* running_mean, float* running_var, float momentum, void* stream) {
  EMO_CHECK(M > 0, "bn_stats: empty batch");
  hipStream_t s = (hipStream_t)stream;
  hipMemsetAsync(mean, 0, sizeof(float) * C, s);
  hipMemsetAsync(var, 0, sizeof(float) * C, s);
  dim3 grid(cdiv(C, 256), cdiv(M, BN_STAT_ROWS));
  EMO_DISPATCH(dtype, (bn_sum_kernel<T><<<grid, 256, 0, s>>>(M, C, (const T*)y, mean)));
  EMO_DISPATCH(dtype, (bn_var_kernel<T><<<grid, 256, 0, s>>>(M, C, (const T*)y, mean, var)));
  bn_finalize_kernel<<<cdiv(C, 256), 256, 0, s>>>(M, C, mean, var, running_mean, running_var, momentum);
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_bn_swish_fwd(int dtype, int M, int C, const void* y, const float* mean,
                                   const float* var, const float* gamma, const float* beta, float eps,
                                   void* z, void* stream) {
  const long n = (long)M * C;
  if (n == 0) return 0;
  if (C % 8 == 0 && (reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(z)) % 16 == 0) {
    EMO_DISPATCH(dtype, (bn_swish_fwd8_kernel<T><<<ew_grid(n / 8), 256, 0, (hipStream_t)stream>>>(
                            n / 8, C, (const T*)y, mean, var, gamma, beta, eps, (T*)z, RowSegs{})));
  } else {
    EMO_DISPATCH(dtype, (bn_swish_fwd_kernel<T><<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(
                            n, C, (const T*)y, mean, var, gamma, beta, eps, (T*)z, RowSegs{})));
  }
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" long emoasr_bn_swish_bwd_scratch_floats(int M, int C) { return ((long)cdiv(M, BN_SUM_ROWS) + 1) * 2 * C; }

// Passes 1 and 2 only (partial sums + fold): leaves tot[2][C] = (mean(dbn), mean(dbn * xhat)) at
// scratch + cdiv(M, BN_SUM_ROWS) * 2 * C and accumulates dgamma / dbeta; the apply pass is then part of emoasr_conv_bwd_fused.
extern "C" int emoasr_bn_swish_bwd_sums(int dtype, int M, int C, const void* dz, const void* y, const float* mean,
                                        const float* var, const float* gamma, const float* beta, float eps,
                                        float* dgamma, float* dbeta, float* scratch, float** tot_out, void* stream) {
  EMO_CHECK(M > 0, "bn_swish_bwd_sums: empty batch");
  hipStream_t s = (hipStream_t)stream;
  EMO_CHECK(C % 8 == 0, "bn_swish_bwd_sums: C=%d must be a multiple of 8", C);
  EMO_CHECK((long)M * C * (dtype == EMO_BF16 ? 2 : 4) < (1L << 32), "bn_swish_bwd_sums: activation larger than 4 GiB");
  const int npart = cdiv(M, BN_SUM_ROWS);
  dim3 sgrid(cdiv(C, 256), npart);
  EMO_DISPATCH(dtype, (bn_bwd_sums_kernel<T><<<sgrid, 256, 0, s>>>(M, C, (const T*)dz, (const T*)y, mean,
                                                                  var, gamma, beta, eps, scratch, RowSegs{})));
  float* tot = scratch + (long)npart * 2 * C;
  EMO_BN_TICKETS(s);
  bn_bwd_fold_kernel<<<cdiv(C, 16), 1024, 0, s>>>(npart, C, 1.f / M, scratch, tot, dgamma, dbeta, RowSegs{}, tickets_);
  if (tot_out) *tot_out = tot;
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_bn_swish_bwd(int dtype, int M, int C, const void* dz, const void* y,
                                   const float* mean, const float* var, const float* gamma,
                                   const float* beta, float eps, void* dy, float* dgamma, float* dbeta,
                                   float* scratch, void* stream) {
  EMO_CHECK(M > 0, "bn_swish_bwd: empty batch");
  hipStream_t s = (hipStream_t)stream;
  EMO_CHECK(C % 8 == 0, "bn_swish_bwd: C=%d must be a multiple of 8", C);
  EMO_CHECK((long)M * C * (dtype == EMO_BF16 ? 2 : 4) < (1L << 32), "bn_swish_bwd: activation larger than 4 GiB");
  const int npart = cdiv(M, BN_SUM_ROWS);
  dim3 grid(cdiv(C, 256), cdiv(M, BN_APPLY_ROWS));
  dim3 sgrid(cdiv(C, 256), npart);
  EMO_DISPATCH(dtype, (bn_bwd_sums_kernel<T><<<sgrid, 256, 0, s>>>(M, C, (const T*)dz, (const T*)y, mean,
                                                                  var, gamma, beta, eps, scratch, RowSegs{})));
  float* tot = scratch + (long)npart * 2 * C;
  EMO_BN_TICKETS(s);
  bn_bwd_fold_kernel<<<cdiv(C, 16), 1024, 0, s>>>(npart, C, 1.f / M, scratch, tot, dgamma, dbeta, RowSegs{}, tickets_);
  EMO_DISPATCH(dtype, (bn_bwd_apply_kernel<T><<<grid, 256, 0, s>>>(M, C, (const T*)dz, (const T*)y, mean,
                                                                  var, gamma, beta, eps, tot, (T*)dy, RowSegs{})));
  EMO_LAUNCH_CHECK();
  return 0;
}

// ---- stacked micro-batches: the BatchNorm kernels over all segments in one launch each (called by csrc/convfused.hip) -----------
int emo_bn_stats_finalize_seg(const RowSegs& sg, int C, const float* part, float* mean, float* var, float* running_mean,
                              float* running_var, float momentum, long long* nbt, hipStream_t s) {
  EMO_CHECK(cdiv(C, BNF_CH) <= 64, "bn_stats_finalize: C=%d too wide for the ticket table", C);
  EMO_BN_TICKETS(s);
  bn_stats_finalize_kernel<<<dim3(cdiv(C, BNF_CH), sg.n > 1 ? sg.n : 1), 1024, 0, s>>>(sg.b0[1], sg.T[0], C, part, mean, var,
                                                                                     running_mean, running_var, momentum, nbt, sg, tickets_);
  EMO_LAUNCH_CHECK();
  return 0;
}

int emo_bn_swish_fwd_seg_dt(int dtype, const RowSegs& sg, int C, const void* y, const float* mean, const float* var,
                            const float* gamma, const float* beta, float eps, void* z, hipStream_t s) {
  const long n = sg.row[sg.n] * C;
  if (n == 0) return 0;
  if (C % 8 == 0) {
    EMO_DISPATCH(dtype, (bn_swish_fwd8_kernel<T><<<ew_grid(n / 8), 256, 0, s>>>(n / 8, C, (const T*)y, mean, var, gamma, beta, eps, (T*)z, sg)));
  } else {
    EMO_DISPATCH(dtype, (bn_swish_fwd_kernel<T><<<ew_grid(n), 256, 0, s>>>(n, C, (const T*)y, mean, var, gamma, beta, eps, (T*)z, sg)));
  }
  EMO_LAUNCH_CHECK();
  return 0;
}
int emo_bn_swish_fwd_seg(const RowSegs& sg, int C, const void* y, const float* mean, const float* var, const float* gamma,
                         const float* beta, float eps, void* z, hipStream_t s) {
  return emo_bn_swish_fwd_seg_dt(EMO_BF16, sg, C, y, mean, var, gamma, beta, eps, z, s);
}
// the apply pass of emoasr_bn_swish_bwd alone (tot = the two means of emoasr_bn_swish_bwd_sums / emo_bn_swish_bwd_sums_seg_dt)
int emo_bn_bwd_apply(int dtype, int M, int C, const void* dz, const void* y, const float* mean, const float* var, const float* gamma,
                     const float* beta, float eps, const float* tot, void* dy, hipStream_t s) {
  dim3 grid(cdiv(C, 256), cdiv(M, BN_APPLY_ROWS));
  EMO_DISPATCH(dtype, (bn_bwd_apply_kernel<T><<<grid, 256, 0, s>>>(M, C, (const T*)dz, (const T*)y, mean, var, gamma, beta, eps, tot, (T*)dy, RowSegs{})));
  EMO_LAUNCH_CHECK();
  return 0;
}
// ... and over all stacked micro-batches in one launch (mean / var [n, C], tot [n, 2, C])
int emo_bn_bwd_apply_seg(int dtype, const RowSegs& sg, int C, const void* dz, const void* y, const float* mean, const float* var,
                         const float* gamma, const float* beta, float eps, const float* tot, void* dy, hipStream_t s) {
  long mmax = 0;
  for (int i = 0; i < sg.n; ++i) mmax = std::max(mmax, sg.row[i + 1] - sg.row[i]);
  dim3 grid(cdiv(C, 256), cdiv((int)mmax, BN_APPLY_ROWS), sg.n);
  EMO_DISPATCH(dtype, (bn_bwd_apply_kernel<T><<<grid, 256, 0, s>>>((int)mmax, C, (const T*)dz, (const T*)y, mean, var, gamma, beta, eps,
                                                                  tot, (T*)dy, sg)));
  EMO_LAUNCH_CHECK();
  return 0;
}
// the depthwise convolution (flip = 0: forward, with the BatchNorm partial statistics when part != NULL; flip = 1: data gradient)
// and its weight gradient over all stacked micro-batches in one launch each (the separate kernels: any dtype)
int emo_dwconv_seg(int dtype, const RowSegs& sg, int tmax, int C, int K, const void* x, const float* w, const float* bias, void* y,
                   int flip, float* part, hipStream_t s) {
  EMO_CHECK(K <= DW_MAXK && (K & 1), "dwconv: K=%d unsupported (odd, <= %d)", K, DW_MAXK);
  if (sg.b0[sg.n] == 0 || tmax == 0) return 0;
  dim3 grid(cdiv(tmax, DW_TT), cdiv(C, 256), sg.b0[sg.n]);
  EMO_DISPATCH(dtype, (dwconv_kernel<T><<<grid, 256, 0, s>>>(tmax, C, K, (const T*)x, w, bias, (T*)y, flip, part, sg)));
  EMO_LAUNCH_CHECK();
  return 0;
}
int emo_dwconv_bwd_w_seg(int dtype, const RowSegs& sg, int tmax, int C, int K, const void* dy, const void* x, float* dw, float* dbias,
                         float* scratch, hipStream_t s) {
  EMO_CHECK(K <= DW_MAXK && (K & 1), "dwconv: K=%d unsupported", K);
  if (sg.b0[sg.n] == 0 || tmax == 0) return 0;
  EMO_CHECK(scratch != nullptr, "dwconv_bwd_w_seg: scratch of emoasr_dwconv_bwd_w_scratch_floats(all utterances, longest) floats required");
  dim3 grid(cdiv(tmax, DW_TT * DW_WCHUNKS), cdiv(C, 256), sg.b0[sg.n]);
  EMO_DISPATCH(dtype, (dwconv_bwd_w_kernel<T><<<grid, 256, 0, s>>>(tmax, C, K, (const T*)dy, (const T*)x, scratch, sg)));
  const int nblk = grid.x * sg.b0[sg.n];
  dwconv_bwd_w_reduce_kernel<<<dim3(cdiv((K + 1) * C, 64), nblk >= 1024 ? 4 : 1), 256, 0, s>>>(nblk, C, K, scratch, dw, dbias);
  EMO_LAUNCH_CHECK();
  return 0;
}

// scratch: [sum over segments of cdiv(M_s, BN_SUM_ROWS) partial rows][2][C], then the means tot [n][2][C] (returned in *tot_out), then
// the segments' raw sums [n][2][C]
int emo_bn_swish_bwd_sums_seg_dt(int dtype, const RowSegs& sg, int C, const void* dz, const void* y, const float* mean, const float* var,
                                 const float* gamma, const float* beta, float eps, float* dgamma, float* dbeta, float* scratch,
                                 float** tot_out, hipStream_t s);
int emo_bn_swish_bwd_sums_seg(const RowSegs& sg, int C, const void* dz, const void* y, const float* mean, const float* var,
                              const float* gamma, const float* beta, float eps, float* dgamma, float* dbeta, float* scratch,
                              float** tot_out, hipStream_t s) {
  return emo_bn_swish_bwd_sums_seg_dt(EMO_BF16, sg, C, dz, y, mean, var, gamma, beta, eps, dgamma, dbeta, scratch, tot_out, s);
}
int emo_bn_swish_bwd_sums_seg_dt(int dtype, const RowSegs& sg, int C, const void* dz, const void* y, const float* mean, const float* var,
                                 const float* gamma, const float* beta, float eps, float* dgamma, float* dbeta, float* scratch,
                                 float** tot_out, hipStream_t s) {
  long mmax = 0;
  for (int i = 0; i < sg.n; ++i) mmax = std::max(mmax, sg.row[i + 1] - sg.row[i]);
  EMO_CHECK(sg.row[sg.n] * C * (dtype == EMO_BF16 ? 2 : 4) < (1L << 32), "bn_swish_bwd_sums_seg: activation larger than 4 GiB");
  dim3 sgrid(cdiv(C, 256), cdiv((int)mmax, BN_SUM_ROWS), sg.n);
  EMO_DISPATCH(dtype, (bn_bwd_sums_kernel<T><<<sgrid, 256, 0, s>>>((int)mmax, C, (const T*)dz, (const T*)y, mean, var, gamma, beta, eps,
                                                                  scratch, sg)));
  float* tot = scratch + sg.sums[sg.n] * 2 * C;
  EMO_CHECK(cdiv(C, 16) <= 64, "bn_swish_bwd_sums_seg: C=%d too wide for the ticket table", C);
  EMO_BN_TICKETS(s);
  bn_bwd_fold_kernel<<<dim3(cdiv(C, 16), sg.n > 1 ? sg.n : 1), 1024, 0, s>>>((int)sg.sums[1], C, 1.f / (float)sg.row[1], scratch, tot,
                                                                         dgamma, dbeta, sg, tickets_);
  if (tot_out) *tot_out = tot;
  EMO_LAUNCH_CHECK();
  return 0;
}
