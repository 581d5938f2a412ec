// Fused kernels of the Conformer convolution module (asr/modeling/conformer.py:98-143), bf16, channels-last [B, T, C]:
//
//   forward   GLU -> depthwise Conv1d(k <= 31) -> BatchNorm partial statistics                  (was 2 launches)
//   backward  BatchNorm/Swish input gradient -> depthwise data gradient -> GLU backward,
//             + the depthwise weight-gradient partials of the same tile                          (was 4 launches)
//
// The element-wise kernels they replace moved every intermediate through HBM ([M, C] each: GLU output, BatchNorm input
// gradient, depthwise data gradient).  Here a workgroup stages the time tile plus its 15-row halo once with 16-byte loads
// (computing the upstream element-wise function on the way into LDS), reads the depthwise weights coalesced through LDS
// (per-thread reads of w[ch][j] were 64 cache lines per wave-instruction: 13.0 -> 9.0 us for the data gradient alone), and
// every thread then takes the window of its two channels from LDS.  The arithmetic, its order and every rounding to bf16
// are those of the separate kernels (convmodule.hip, elementwise.hip), so the results are bit-identical to the unfused
// sequence; the weight-gradient partials keep their layout and are folded by the same reduce kernel.
// Measured (MI355X, B 22 x T' 320 x 256, HIP-graph timed): forward 19.0 us against 22.0 (GLU 5.4 + convolution 16.6 incl.
// the statistics merge); backward 56.6 us against 52.4 for the separate launches (11 + 6 us of BatchNorm sums and fold and
// 6 us of weight-gradient fold are common to both): the fused backward kernel (~33 us) is bound by its serial phases at
// one workgroup per CU (62 x 8 sigmoids per thread in the staging pass, quarter-rate exp / rcp), not by memory.  The
// training step is unchanged within noise (9.69 vs 9.71 ms); the fused path is the default because it needs three [M, C]
// buffers and five launches per layer less.
#include <algorithm>
#include "common.h"
#include "../../include/emoasr_hip.h"

namespace {

constexpr int CF_TT = 32;     // output frames per workgroup (= DW_TT of convmodule.hip: same BatchNorm / weight-gradient partial tables)
constexpr int CF_MAXK = 31;
constexpr int CF_ROWS = CF_TT + CF_MAXK - 1;  // 62 staged rows
constexpr int CF_CB = 256;                     // channels (= threads) per workgroup (128: 2-3 workgroups per CU, measured
                                              // 10 % slower: twice the weight staging and barriers per channel)
constexpr int CF_ROWS_P = 64;                 // LDS rows (the staging passes are 8 full sweeps of 8 rows: no tail branch)

// Work split of the compute phases (both kernels): a thread owns TWO adjacent channels (packed f32x2 multiplies and adds:
// half the VALU instructions of one channel per thread) and one HALF of the tile -- 16 of the 32 output frames of a
// convolution, or 16 of the (up to) 31 taps of the weight gradient, whose sums over the 32 frames then keep the frame order
// of the unfused kernel.  Every tap is ONE fused multiply-add (emo_mac2 of common.h, v_pk_fma_f32), as in the kernels these
// replace (emo_mac): round 4 -- the two kernels are VALU-bound, the stencils were half of their instructions.
typedef __attribute__((ext_vector_type(2))) float f2;
__device__ __forceinline__ f2 unpack2(unsigned u) { return f2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; }
__device__ __forceinline__ unsigned pack2(f2 v) {
  const bf16 lo = (bf16)v[0], hi = (bf16)v[1];
  return (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
}
__device__ __forceinline__ unsigned lds_pair(const bf16* base, int row, int pair) {
  return *reinterpret_cast<const unsigned*>(base + row * CF_CB + 2 * pair);
}

// ---- forward: c[b,t,ch] = bias[ch] + sum_j w[ch,j] * z[b, t + j - pad, ch],  z = GLU(g) = g[:, :C] * sigmoid(g[:, C:]) ----
// GLU = false: x is the [B, T, C] input itself (the plain depthwise convolution; flip = 1 gives the data gradient).
template <bool GLU>
__global__ __launch_bounds__(CF_CB) void cf_dwconv_kernel(int Tn, int C, int K, const bf16* __restrict__ x,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        bf16* __restrict__ y, int flip, float* __restrict__ part,
                                                        const RowSegs sg) {
  __shared__ __attribute__((aligned(16))) bf16 zs[CF_ROWS_P * CF_CB];
  __shared__ __attribute__((aligned(16))) bf16 os[CF_TT * CF_CB];    // the output tile (stored rows, BatchNorm partials)
  const int tid = threadIdx.x;
  const int cb = blockIdx.y * CF_CB;                 // first channel of this workgroup
  const int ldx = GLU ? 2 * C : C;
  int b = blockIdx.z, nx = gridDim.x;
  if (sg.n > 1) {   // stacked micro-batches: this utterance's segment (own padded length, own partial-statistics table)
    const int si = rowsegs_of_utt(sg, b);
    Tn = sg.T[si];
    x += sg.row[si] * ldx;
    y += sg.row[si] * C;
    if (part) part += sg.part[si];
    b -= sg.b0[si];
    nx = (Tn + CF_TT - 1) / CF_TT;
    if ((int)blockIdx.x >= nx) return;   // (the grid follows the longest segment)
  }
  const int t0 = blockIdx.x * CF_TT, pad = (K - 1) / 2;
  const int nch = min(CF_CB, C - cb);                // channels present (multiple of 8)
  // ---- this thread's taps: w[ch][K] is read once, coalesced, through LDS (per-thread reads of w[ch * K + j] are 64
  //      different cache lines per wave-instruction: 62 such gathers per thread cost more than the convolution) ----
  const int pr = tid & (CF_CB / 2 - 1), half = tid / (CF_CB / 2);
  f2 wr[CF_MAXK];
  {
    float* wsm = reinterpret_cast<float*>(zs);   // CF_CB * K floats <= the bytes of zs
    for (int idx = tid; idx < nch * K; idx += CF_CB) wsm[idx] = w[(long)cb * K + idx];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CF_MAXK; ++j) {
      const int jj = flip ? K - 1 - j : j;
      wr[j] = (j < K && 2 * pr < nch) ? f2{wsm[(2 * pr) * K + jj], wsm[(2 * pr + 1) * K + jj]} : f2{0.f, 0.f};
    }
    __syncthreads();
  }
  // ---- stage rows t0 - pad .. t0 + TT - 1 + pad (zeros outside the utterance: Conv1d's zero padding) ----
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(x + (long)b * Tn * ldx);
  const int ch8 = (tid & (CF_CB / 8 - 1)) * 8;
  {
    // all 8 (16) loads of a thread are issued before the first is used: one memory round trip for the whole tile
    Vec16<bf16> sa[CF_ROWS_P / 8], sg[GLU ? CF_ROWS_P / 8 : 1];
#pragma unroll
    for (int it = 0; it < CF_ROWS_P / 8; ++it) {
      const int row = tid / (CF_CB / 8) + it * 8;
      const int t = t0 + row - pad;
      const bool ok = row < CF_TT + K - 1 && t >= 0 && t < Tn && ch8 < nch;
      const unsigned off = ok ? (unsigned)(((long)t * ldx + cb + ch8) * 2) : EMO_OOB;
      sa[it] = buf_load16<bf16>(rs, off);
      if constexpr (GLU) sg[it] = buf_load16<bf16>(rs, ok ? off + (unsigned)(C * 2) : EMO_OOB);
    }
#pragma unroll
    for (int it = 0; it < CF_ROWS_P / 8; ++it) {
      const int row = tid / (CF_CB / 8) + it * 8;
      Vec16<bf16> a = sa[it];
      if constexpr (GLU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) a.v[e] = (bf16)((float)a.v[e] * sigmoidf_((float)sg[it].v[e]));
      }
      store16(&zs[row * CF_CB + ch8], a);
    }
  }
  __syncthreads();
  // ---- 16 output frames x 2 channels per thread ----
  {
    const int c0 = cb + 2 * pr;
    if (2 * pr < nch) {
      const f2 bv = bias ? f2{bias[c0], bias[c0 + 1]} : f2{0.f, 0.f};
      f2 win[CF_TT / 2 + CF_MAXK - 1];
#pragma unroll
      for (int i = 0; i < CF_TT / 2 + CF_MAXK - 1; ++i) win[i] = unpack2(lds_pair(zs, half * (CF_TT / 2) + i, pr));
#pragma unroll
      for (int i = 0; i < CF_TT / 2; ++i) {
        f2 acc = bv;
#pragma unroll
        for (int j = 0; j < CF_MAXK; ++j) acc = emo_mac2(wr[j], win[i + j], acc);
        *reinterpret_cast<unsigned*>(os + (half * (CF_TT / 2) + i) * CF_CB + 2 * pr) = pack2(acc);
      }
    }
  }
  __syncthreads();
  // ---- full-row stores of the tile; BatchNorm partials per channel in frame order ----
#pragma unroll
  for (int it = 0; it < CF_TT / 8; ++it) {
    const int row = tid / (CF_CB / 8) + it * 8;
    if (t0 + row < Tn && ch8 < nch)
      *reinterpret_cast<bf16x8*>(y + ((long)b * Tn + t0 + row) * C + cb + ch8) = *reinterpret_cast<const bf16x8*>(os + row * CF_CB + ch8);
  }
  if (part && tid < nch) {
    float out[CF_TT];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CF_TT; ++i) {
      out[i] = t0 + i < Tn ? (float)os[i * CF_CB + tid] : 0.f;
      s += out[i];
    }
    const int n = min(CF_TT, Tn - t0);
    const float mb = s / n;
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < CF_TT; ++i) {
      const float d = out[i] - mb;
      m2 += i < n ? d * d : 0.f;
    }
    float* p = part + ((long)b * nx + blockIdx.x) * 2 * C + cb + tid;
    p[0] = s;
    p[C] = m2;
  }
}

// ---- backward ---------------------------------------------------------------------------------------------------
//   dbn = ds * swish'(gamma * xhat + beta),  dc = gamma * invstd * (dbn - mean(dbn) - xhat * mean(dbn * xhat))   (bf16)
//   dz[t] = sum_j w[K-1-j] * dc[t + j - pad]                                                                      (bf16)
//   dg[:, :C] = dz * sigmoid(gb),  dg[:, C:] = dz * ga * sigmoid(gb) * (1 - sigmoid(gb))
//   wpart[blk][j][ch] = sum_{t in tile} dc[t] * z[t + j - pad],  wpart[blk][K][ch] = sum dc     (z = GLU(g), bf16)
// tot: [2][C] means from the BatchNorm fold (emoasr_bn_swish_bwd_sums).
__global__ __launch_bounds__(CF_CB) void cf_conv_bwd_kernel(int Tn, int C, int K, const bf16* __restrict__ ds,
                                                          const bf16* __restrict__ cv, const float* __restrict__ mean,
                                                          const float* __restrict__ var, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps,
                                                          const float* __restrict__ tot, const bf16* __restrict__ g,
                                                          const float* __restrict__ w, bf16* __restrict__ dg,
                                                          float* __restrict__ wpart, const RowSegs sg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* dcs = reinterpret_cast<bf16*>(smem);                       // [64][CB]  BatchNorm input gradient (tile + halo)
  bf16* zs = dcs + CF_ROWS_P * CF_CB;                                // [64][CB]  GLU output (tile + halo)
  // (the raw GLU halves of the tile's own frames used to sit in two more [32][CB] images: 96 KB per workgroup, ONE workgroup per
  // CU, every phase's memory round trip exposed.  They are re-read from global memory in the last phase instead -- L2-resident,
  // this workgroup staged them microseconds ago -- which leaves 64 KB and two workgroups per CU.)
  const int tid = threadIdx.x;
  const int cb = blockIdx.y * CF_CB;
  const int nch = min(CF_CB, C - cb);
  int b = blockIdx.z;
  if (sg.n > 1) {   // stacked micro-batches: this utterance's segment (own padded length, batch statistics and means)
    const int si = rowsegs_of_utt(sg, b);
    Tn = sg.T[si];
    ds += sg.row[si] * C; cv += sg.row[si] * C; g += sg.row[si] * 2 * C; dg += sg.row[si] * 2 * C;
    mean += (long)si * C; var += (long)si * C; tot += (long)si * 2 * C;
    b -= sg.b0[si];
    if ((int)blockIdx.x * CF_TT >= Tn) {
      // (the grid follows the longest segment: this block has no frames, but its slot of the weight-gradient partial table is
      // summed by the reduce kernel like every other)
      float* p = wpart + ((long)blockIdx.z * gridDim.x + blockIdx.x) * (K + 1) * C + cb + tid;
      if (tid < nch)
        for (int j = 0; j <= K; ++j) p[(long)j * C] = 0.f;
      return;
    }
  }
  const int t0 = blockIdx.x * CF_TT, pad = (K - 1) / 2;
  const int ch8 = (tid & (CF_CB / 8 - 1)) * 8;
  const bool chok = ch8 < nch;
  const int pr = tid & (CF_CB / 2 - 1), half = tid / (CF_CB / 2);
  f2 wr[CF_MAXK];   // flipped taps of this thread's two channels, read coalesced through LDS (see cf_dwconv_kernel)
  {
    float* wsm = reinterpret_cast<float*>(smem);
    for (int idx = tid; idx < nch * K; idx += CF_CB) wsm[idx] = w[(long)cb * K + idx];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CF_MAXK; ++j)
      wr[j] = (j < K && 2 * pr < nch) ? f2{wsm[(2 * pr) * K + (K - 1 - j)], wsm[(2 * pr + 1) * K + (K - 1 - j)]} : f2{0.f, 0.f};
    __syncthreads();
  }
  {
    const __amdgpu_buffer_rsrc_t rsd = make_rsrc(ds + (long)b * Tn * C), rsc = make_rsrc(cv + (long)b * Tn * C);
    const __amdgpu_buffer_rsrc_t rsg = make_rsrc(g + (long)b * Tn * 2 * C);
    // the GLU input of the staged rows: z into LDS for all of them, the raw halves for the tile's own frames
    {
      Vec16<bf16> ga[CF_ROWS_P / 8], gb[CF_ROWS_P / 8];
#pragma unroll
      for (int it = 0; it < CF_ROWS_P / 8; ++it) {
        const int row = tid / (CF_CB / 8) + it * 8;
        const int t = t0 + row - pad;
        const bool ok = row < CF_TT + K - 1 && t >= 0 && t < Tn && chok;
        const unsigned goff = ok ? (unsigned)(((long)t * 2 * C + cb + ch8) * 2) : EMO_OOB;
        ga[it] = buf_load16<bf16>(rsg, goff);
        gb[it] = buf_load16<bf16>(rsg, ok ? goff + (unsigned)(C * 2) : EMO_OOB);
      }
#pragma unroll
      for (int it = 0; it < CF_ROWS_P / 8; ++it) {
        const int row = tid / (CF_CB / 8) + it * 8;
        const int core = row - pad;
        (void)core;
        Vec16<bf16> z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z.v[e] = (bf16)((float)ga[it].v[e] * sigmoidf_((float)gb[it].v[e]));
        store16(&zs[row * CF_CB + ch8], z);
      }
    }
    // ... then the BatchNorm input gradient of the same rows
    float mu[8], is[8], gm[8], bt[8], m1[8], m2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int cc = cb + ch8 + j;
      mu[j] = chok ? mean[cc] : 0.f; is[j] = chok ? rsqrtf(var[cc] + eps) : 0.f;
      gm[j] = chok ? gamma[cc] : 0.f; bt[j] = chok ? beta[cc] : 0.f;
      m1[j] = chok ? tot[cc] : 0.f; m2[j] = chok ? tot[C + cc] : 0.f;
    }
    Vec16<bf16> dvs[CF_ROWS_P / 8], yvs[CF_ROWS_P / 8];
#pragma unroll
    for (int it = 0; it < CF_ROWS_P / 8; ++it) {
      const int row = tid / (CF_CB / 8) + it * 8;
      const int t = t0 + row - pad;
      const bool ok = row < CF_TT + K - 1 && t >= 0 && t < Tn && chok;
      const unsigned off = ok ? (unsigned)(((long)t * C + cb + ch8) * 2) : EMO_OOB;
      dvs[it] = buf_load16<bf16>(rsd, off);
      yvs[it] = buf_load16<bf16>(rsc, off);
    }
#pragma unroll
    for (int it = 0; it < CF_ROWS_P / 8; ++it) {
      const int row = tid / (CF_CB / 8) + it * 8;
      const int t = t0 + row - pad;
      const bool ok = row < CF_TT + K - 1 && t >= 0 && t < Tn && chok;
      const Vec16<bf16> dv = dvs[it], yv = yvs[it];
      Vec16<bf16> o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = ((float)yv.v[e] - mu[e]) * is[e];
        const float dbn = (float)dv.v[e] * dswishf_(gm[e] * xh + bt[e]);
        o.v[e] = ok ? (bf16)(gm[e] * is[e] * (dbn - m1[e] - xh * m2[e])) : (bf16)0.f;
      }
      store16(&dcs[row * CF_CB + ch8], o);
    }
  }
  __syncthreads();
  if (2 * pr >= nch) return;
  const int c0 = cb + 2 * pr;
  // ---- data gradient of the depthwise convolution + GLU backward: 16 frames x 2 channels ----
  {
    f2 dcw[CF_TT / 2 + CF_MAXK - 1];
#pragma unroll
    for (int i = 0; i < CF_TT / 2 + CF_MAXK - 1; ++i) dcw[i] = unpack2(lds_pair(dcs, half * (CF_TT / 2) + i, pr));
    bf16* dgb = dg + (long)b * Tn * 2 * C + c0;
#pragma unroll
    for (int i = 0; i < CF_TT / 2; ++i) {
      f2 acc = f2{0.f, 0.f};
#pragma unroll
      for (int j = 0; j < CF_MAXK; ++j) acc = emo_mac2(wr[j], dcw[i + j], acc);
      const f2 d = unpack2(pack2(acc));   // the data gradient is stored as bf16 by the unfused kernel
      const int fr = half * (CF_TT / 2) + i;
      const bool fok = t0 + fr < Tn;
      const bf16* gp = g + ((long)b * Tn + (fok ? t0 + fr : 0)) * 2 * C + c0;
      const f2 a = fok ? unpack2(*reinterpret_cast<const unsigned*>(gp)) : f2{0.f, 0.f};
      const f2 gt = fok ? unpack2(*reinterpret_cast<const unsigned*>(gp + C)) : f2{0.f, 0.f};
      const f2 sg = f2{sigmoidf_(gt[0]), sigmoidf_(gt[1])};
      if (t0 + fr < Tn) {
        *reinterpret_cast<unsigned*>(dgb + (long)(t0 + fr) * 2 * C) = pack2(d * sg);
        *reinterpret_cast<unsigned*>(dgb + (long)(t0 + fr) * 2 * C + C) = pack2(d * a * sg * (f2{1.f, 1.f} - sg));
      }
    }
  }
  // ---- weight-gradient partials: 16 taps x 2 channels, summed over the tile's 32 frames in frame order ----
  {
    constexpr int TAPS = (CF_MAXK + 1) / 2;   // 16
    const int j0 = half * TAPS;
    f2 dcc[CF_TT];
#pragma unroll
    for (int i = 0; i < CF_TT; ++i) dcc[i] = unpack2(lds_pair(dcs, i + pad, pr));   // zero past the utterance's end
    f2 zw[CF_TT + TAPS - 1];
#pragma unroll
    for (int i = 0; i < CF_TT + TAPS - 1; ++i) zw[i] = j0 + i < CF_ROWS_P ? unpack2(lds_pair(zs, j0 + i, pr)) : f2{0.f, 0.f};
    f2 acc[TAPS];
#pragma unroll
    for (int j = 0; j < TAPS; ++j) acc[j] = f2{0.f, 0.f};
    f2 sb = f2{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < CF_TT; ++i) {
      const f2 d = dcc[i];
      sb += d;
#pragma unroll
      for (int j = 0; j < TAPS; ++j) acc[j] = emo_mac2(d, zw[i + j], acc[j]);
    }
    const long blk = ((long)blockIdx.z * gridDim.x + blockIdx.x);
    float* p = wpart + blk * (K + 1) * C + c0;
#pragma unroll
    for (int j = 0; j < TAPS; ++j)
      if (j0 + j < K) *reinterpret_cast<f2*>(p + (long)(j0 + j) * C) = acc[j];
    if (half == 0) *reinterpret_cast<f2*>(p + (long)K * C) = sb;
  }
}

}  // namespace

int emo_dwconv_bwd_w_reduce(int nblk, int C, int K, const float* part, float* dw, float* dbias, hipStream_t s);  // convmodule.hip

// c = depthwise_conv(GLU(g)) (+ per-block BatchNorm partial statistics when part != NULL, consumed by
// emoasr_bn_stats_finalize): conformer.py:126-131 without materialising the GLU output.  g: [B*T, 2C].
extern "C" int emoasr_glu_dwconv_fwd(int dtype, int B, int Tn, int C, int K, const void* g, const float* w,
                                     const float* bias, void* c, float* part, void* stream) {
  EMO_CHECK(dtype == EMO_BF16, "glu_dwconv_fwd: bf16 only");
  EMO_CHECK(K <= CF_MAXK && (K & 1) && C % 8 == 0, "glu_dwconv_fwd: K=%d (odd, <= %d), C=%d (multiple of 8)", K, CF_MAXK, C);
  EMO_CHECK((long)Tn * 2 * C * 2 < (1L << 32), "glu_dwconv_fwd: utterance larger than 4 GiB");
  if (B * Tn == 0) return 0;
  dim3 grid(cdiv(Tn, CF_TT), cdiv(C, CF_CB), B);
  cf_dwconv_kernel<true><<<grid, CF_CB, 0, (hipStream_t)stream>>>(Tn, C, K, (const bf16*)g, w, bias, (bf16*)c, 0, part, RowSegs{});
  EMO_LAUNCH_CHECK();
  return 0;
}

// The LDS-staged depthwise convolution on a plain [B, T, C] input (flip = 1: its data gradient); called by
// emoasr_dwconv_fwd / _fwd_stats / _bwd_x for bf16.
int emo_dwconv_lds(int B, int Tn, int C, int K, const void* x, const float* w, const float* bias, void* y, int flip,
                   float* part, hipStream_t s) {
  dim3 grid(cdiv(Tn, CF_TT), cdiv(C, CF_CB), B);
  cf_dwconv_kernel<false><<<grid, CF_CB, 0, s>>>(Tn, C, K, (const bf16*)x, w, bias, (bf16*)y, flip, part, RowSegs{});
  EMO_LAUNCH_CHECK();
  return 0;
}

// Backward of BatchNorm(training) -> Swish ... depthwise conv ... GLU in one launch, given the BatchNorm means of
// emoasr_bn_swish_bwd_sums in `tot`:  ds = gradient w.r.t. the Swish output, cv = the depthwise convolution's output,
// g = the GLU input [B*T, 2C]  ->  dg [B*T, 2C];  dw [C, K] and dbias [C] are accumulated into.
// scratch: emoasr_dwconv_bwd_w_scratch_floats(B, Tn, C, K) floats.
extern "C" int emoasr_conv_bwd_fused(int dtype, int B, int Tn, int C, int K, const void* ds, const void* cv,
                                     const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                                     const float* tot, const void* g, const float* w, void* dg, float* dw, float* dbias,
                                     float* scratch, void* stream) {
  EMO_CHECK(dtype == EMO_BF16, "conv_bwd_fused: bf16 only");
  EMO_CHECK(K <= CF_MAXK && (K & 1) && C % 8 == 0, "conv_bwd_fused: K=%d (odd, <= %d), C=%d (multiple of 8)", K, CF_MAXK, C);
  EMO_CHECK((long)Tn * 2 * C * 2 < (1L << 32), "conv_bwd_fused: utterance larger than 4 GiB");
  EMO_CHECK(scratch != nullptr && tot != nullptr, "conv_bwd_fused: scratch and tot are required");
  if (B * Tn == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  constexpr int bytes = 2 * CF_ROWS_P * CF_CB * 2;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)cf_conv_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { emo_set_error("hipFuncSetAttribute(%d): %s", bytes, hipGetErrorString(e)); return 1; }
    attr_done = true;
  }
  dim3 grid(cdiv(Tn, CF_TT), cdiv(C, CF_CB), B);
  cf_conv_bwd_kernel<<<grid, CF_CB, bytes, s>>>(Tn, C, K, (const bf16*)ds, (const bf16*)cv, mean, var, gamma, beta, eps, tot,
                                              (const bf16*)g, w, (bf16*)dg, scratch, RowSegs{});
  EMO_LAUNCH_CHECK();
  return emo_dwconv_bwd_w_reduce(grid.x * B, C, K, scratch, dw, dbias, s);
}

// ---- the convolution module's per-utterance part for STACKED micro-batches (emoasr_segments_t): every kernel takes all segments
// in one launch.  Same arithmetic as calling the entry points above once per segment, in order. -------------------------------
int emo_bn_stats_finalize_seg(const RowSegs& sg, int C, const float* part, float* mean, float* var, float* running_mean,
                              float* running_var, float momentum, long long* nbt, hipStream_t s);                    // convmodule.hip
int emo_bn_swish_fwd_seg(const RowSegs& sg, int C, const void* y, const float* mean, const float* var, const float* gamma,
                         const float* beta, float eps, void* z, hipStream_t s);                                       // convmodule.hip
int emo_bn_swish_bwd_sums_seg(const RowSegs& sg, int C, const void* dz, const void* y, const float* mean, const float* var,
                              const float* gamma, const float* beta, float eps, float* dgamma, float* dbeta, float* scratch,
                              float** tot_out, hipStream_t s);                                                        // convmodule.hip

namespace {
bool row_segs(const emoasr_segments_t* seg, int C, RowSegs* out, int* tmax) {
  if (!seg || seg->n < 1 || seg->n > EMOASR_MAX_SEGMENTS) return false;
  RowSegs sg{};
  sg.n = seg->n;
  *tmax = 0;
  for (int i = 0; i < seg->n; ++i) {
    if (seg->B[i] <= 0 || seg->T[i] <= 0) return false;
    sg.b0[i + 1] = sg.b0[i] + seg->B[i];
    sg.T[i] = seg->T[i];
    sg.row[i + 1] = sg.row[i] + (long)seg->B[i] * seg->T[i];
    sg.part[i + 1] = sg.part[i] + emoasr_dwconv_stats_floats(seg->B[i], seg->T[i], C);
    sg.sums[i + 1] = sg.sums[i] + (emoasr_bn_swish_bwd_scratch_floats(seg->B[i] * seg->T[i], C) / (2 * C) - 1);
    *tmax = std::max(*tmax, seg->T[i]);
  }
  *out = sg;
  return true;
}
}  // namespace

// c = depthwise_conv(GLU(g)) per segment, BatchNorm batch statistics per segment (training: bmean / bvar [n, C], the running
// statistics updated once per segment in order; eval: the running statistics), z = Swish(BatchNorm(c)).  g [M, 2C], c / z [M, C],
// part: sum over the segments of emoasr_dwconv_stats_floats(B[s], T[s], C) floats.  conformer.py:126-133.
extern "C" int emoasr_conv_module_fwd_seg(int dtype, const emoasr_segments_t* seg, int C, int K, const void* g, const float* w,
                                          const float* bias, void* c, float* part, float* bmean, float* bvar, float* running_mean,
                                          float* running_var, float momentum, long long* num_batches_tracked, const float* gamma,
                                          const float* beta, float eps, void* z, int training, void* stream) {
  EMO_CHECK(dtype == EMO_BF16, "conv_module_fwd_seg: bf16 only");
  EMO_CHECK(K <= CF_MAXK && (K & 1) && C % 8 == 0, "conv_module_fwd_seg: K=%d (odd, <= %d), C=%d (multiple of 8)", K, CF_MAXK, C);
  RowSegs sg;
  int tmax = 0;
  EMO_CHECK(row_segs(seg, C, &sg, &tmax), "conv_module_fwd_seg: bad segment description");
  EMO_CHECK((long)tmax * 2 * C * 2 < (1L << 32) && sg.row[sg.n] * 2 * C * 2 < (1L << 46), "conv_module_fwd_seg: batch too large");
  EMO_CHECK(!training || (part && bmean && bvar), "conv_module_fwd_seg: training needs the BatchNorm buffers");
  hipStream_t s = (hipStream_t)stream;
  // algorithmic bytes per row and channel: g (2) in, c out; c in, z out
  EmoTimerScope timer_(EMO_TIMER_CONV_MODULE, s, 0.0, 5.0 * (double)sg.row[sg.n] * C * 2.0);
  dim3 grid(cdiv(tmax, CF_TT), cdiv(C, CF_CB), sg.b0[sg.n]);
  cf_dwconv_kernel<true><<<grid, CF_CB, 0, s>>>(tmax, C, K, (const bf16*)g, w, bias, (bf16*)c, 0, training ? part : nullptr, sg);
  EMO_LAUNCH_CHECK();
  if (training) {
    if (emo_bn_stats_finalize_seg(sg, C, part, bmean, bvar, running_mean, running_var, momentum, num_batches_tracked, s)) return 1;
    return emo_bn_swish_fwd_seg(sg, C, c, bmean, bvar, gamma, beta, eps, z, s);
  }
  RowSegs one{};   // eval: the same (running) statistics for every row
  one.n = 1; one.b0[1] = sg.b0[sg.n]; one.row[1] = sg.row[sg.n];
  return emo_bn_swish_fwd_seg(one, C, c, running_mean, running_var, gamma, beta, eps, z, s);
}

// scratch of emoasr_conv_module_bwd_seg: which = 0 the BatchNorm partial sums + means (floats), 1 the depthwise weight-gradient
// partials (floats)
extern "C" long emoasr_conv_module_bwd_seg_scratch_floats(const emoasr_segments_t* seg, int C, int K, int which) {
  RowSegs sg;
  int tmax = 0;
  if (!row_segs(seg, C, &sg, &tmax)) return 0;
  if (which == 0) return (sg.sums[sg.n] + 2 * sg.n) * 2 * C;   // partial rows, the means, the raw sums
  return emoasr_dwconv_bwd_w_scratch_floats(sg.b0[sg.n], tmax, C, K);
}

// dz (gradient w.r.t. the Swish output) -> dg [M, 2C] through BatchNorm(training, per-segment statistics) / Swish, the depthwise
// convolution and the GLU; dgamma / dbeta / dw / dbias accumulated.  Autograd of conformer.py:126-133.
extern "C" int emoasr_conv_module_bwd_seg(int dtype, const emoasr_segments_t* seg, int C, int K, const void* dz, const void* c,
                                          const float* bmean, const float* bvar, const float* gamma, const float* beta, float eps,
                                          float* dgamma, float* dbeta, const void* g, const float* w, void* dg, float* dw,
                                          float* dbias, float* bn_scratch, float* dw_scratch, void* stream) {
  EMO_CHECK(dtype == EMO_BF16, "conv_module_bwd_seg: bf16 only");
  EMO_CHECK(K <= CF_MAXK && (K & 1) && C % 8 == 0, "conv_module_bwd_seg: K=%d (odd, <= %d), C=%d (multiple of 8)", K, CF_MAXK, C);
  EMO_CHECK(bn_scratch && dw_scratch, "conv_module_bwd_seg: scratch required");
  RowSegs sg;
  int tmax = 0;
  EMO_CHECK(row_segs(seg, C, &sg, &tmax), "conv_module_bwd_seg: bad segment description");
  hipStream_t s = (hipStream_t)stream;
  // algorithmic bytes per row and channel: dz, c in (sums); dz, c, g (2) in, dg (2) out
  EmoTimerScope timer_(EMO_TIMER_CONV_MODULE, s, 0.0, 8.0 * (double)sg.row[sg.n] * C * 2.0);
  float* tot = nullptr;
  if (emo_bn_swish_bwd_sums_seg(sg, C, dz, c, bmean, bvar, gamma, beta, eps, dgamma, dbeta, bn_scratch, &tot, s)) return 1;
  constexpr int bytes = 2 * CF_ROWS_P * CF_CB * 2;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)cf_conv_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { emo_set_error("hipFuncSetAttribute(%d): %s", bytes, hipGetErrorString(e)); return 1; }
    attr_done = true;
  }
  dim3 grid(cdiv(tmax, CF_TT), cdiv(C, CF_CB), sg.b0[sg.n]);
  cf_conv_bwd_kernel<<<grid, CF_CB, bytes, s>>>(tmax, C, K, (const bf16*)dz, (const bf16*)c, bmean, bvar, gamma, beta, eps, tot,
                                              (const bf16*)g, w, (bf16*)dg, dw_scratch, sg);
  EMO_LAUNCH_CHECK();
  return emo_dwconv_bwd_w_reduce(grid.x * sg.b0[sg.n], C, K, dw_scratch, dw, dbias, s);
}
