// One expansion round of the transducer's alignment-length-synchronous beam search (asr/modeling/decoders/rnn_transducer.py:242-325)
// for up to 16 live hypotheses in FIVE launches instead of the 27 small dependent kernels of the launch chain
// (engine._rnnt_beam_search_chain: embedding, two products + a cell kernel per LSTM layer, eight state gathers / scatters, the
// joint's two products, tanh, log-softmax, top-k, three copies -- 127 us per round inside a replayed graph):
//
//   emoasr_rnnt_beam_lstm   x 2   one LSTM layer step for every hypothesis: input rows gathered by index (embedding rows by label
//                                 id / the layer below's new h by destination slot), previous state read from the slot pool at
//                                 src[i], both products, the cell, new state written to the pool at dst[i]
//   emoasr_rnnt_beam_joint        g = w_dec . h + b, joint input tanh(e_t + g) (frame index t read from the device control words)
//   emoasr_gemm_nt                the output layer (unchanged)
//   emoasr_rnnt_beam_pick         log-softmax, the blank log-probability and the beam_width best non-blank labels per hypothesis,
//                                 written straight into the round's result record
//
// The control words (labels, source / destination slots, frame index) stay what engine._rnnt_beam_round_graph uploads; the host
// bookkeeping (stable sort by float64 score, merge of equal label sequences, cut to the beam) is unchanged.
//
// Numerics follow the chain: products accumulate in f32, every intermediate is rounded to the compute dtype where the chain's
// kernels store it (gate pre-activations after each of the two products, h, the joint input); the summation ORDER inside a dot
// product differs from the MFMA kernels', so near-ties of the top-k can differ (tests: the reference's golden hypotheses, both forms).
#include <math.h>
#include "common.h"
#include "../../include/emoasr_hip.h"

namespace {

constexpr int BT = 256;        // threads per workgroup
constexpr int NBMAX = 16;      // hypotheses per round
constexpr int UN = 8;          // hidden units per workgroup (4 UN = 32 gate rows = the 32 eight-lane row groups of a workgroup)
int g_beam_mfma = 1;           // option "rnnt_beam_mfma": bf16 LSTM / joint steps on the matrix cores (0: the VALU kernels, as f32)

template <typename T> __device__ __forceinline__ float rnd(float x) { return to_f32(from_f32<T>(x)); }

// acc[i] += sum_e w[k0 + e] * x_i[k0 + e] for this lane's 16-byte pieces of one weight row, every hypothesis i < nb.
// xs: f32 [NBMAX][ldx] in LDS.  8 lanes share a row (sub = lane's piece index): pieces k0 = sub * VEC, + 8 * VEC, ...
template <typename T>
__device__ __forceinline__ void row_dots(const T* __restrict__ w, int K, const float* __restrict__ xs, int ldx, int nb, int sub,
                                         float (&acc)[NBMAX]) {
  constexpr int VEC = 16 / sizeof(T);
  for (int k = sub * VEC; k < K; k += 8 * VEC) {
    float wv[VEC];
    if constexpr (sizeof(T) == 2) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(w + k);
#pragma unroll
      for (int e = 0; e < 8; ++e) wv[e] = (float)v[e];
    } else {
      const f32x4 v = *reinterpret_cast<const f32x4*>(w + k);
#pragma unroll
      for (int e = 0; e < 4; ++e) wv[e] = v[e];
    }
#pragma unroll
    for (int i = 0; i < NBMAX; ++i) {
      if (i < nb) {
        const float* x = xs + i * ldx + k;
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) s += wv[e] * x[e];
        acc[i] += s;
      }
    }
  }
}
__device__ __forceinline__ float group_sum8(float v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
  return v;
}

struct BeamLstmArgs {
  int nb, nin, H;
  const void* xtab; long ldx;          // input rows: xtab[xidx[i]] (nin values of T)
  const long long* xidx;
  const void *w_ih, *w_hh;             // [4H][nin], [4H][H]
  const float* bias;                   // [4H] = bias_ih + bias_hh
  void* ph; float* pc;                 // state pools [slots][H] (T / f32)
  const long long *src, *dst;
  const long long* copy_src; long long* copy_dst; int copy_n;   // optional: workgroup 0 also copies copy_n words (the control record
                                                                 // from pinned host memory to its device twin, for the later launches)
};

template <typename T>
__global__ __launch_bounds__(BT) void rnnt_beam_lstm_kernel(const BeamLstmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nb = a.nb, nin = a.nin, H = a.H, tid = threadIdx.x, grp = tid >> 3, sub = tid & 7;
  float* xs = reinterpret_cast<float*>(smem);          // [NBMAX][nin]
  float* hs = xs + NBMAX * nin;                        // [NBMAX][H]
  float* gs = hs + NBMAX * H;                          // [NBMAX][4 UN] gate pre-activations of this workgroup's units
  const T* xtab = static_cast<const T*>(a.xtab);
  const T* ph = static_cast<const T*>(a.ph);
  // the control words may live in pinned HOST memory (the search's zero-copy hand-off): read each once, through LDS
  __shared__ long long cw[3][NBMAX];
  if (tid < nb) { cw[0][tid] = a.xidx[tid]; cw[1][tid] = a.src[tid]; cw[2][tid] = a.dst[tid]; }
  if (blockIdx.x == 0 && a.copy_dst && tid >= 64 && tid - 64 < a.copy_n) a.copy_dst[tid - 64] = a.copy_src[tid - 64];
  __syncthreads();
  for (int i = 0; i < nb; ++i) {
    const T* xr = xtab + cw[0][i] * a.ldx;
    const T* hr = ph + cw[1][i] * (long)H;
    for (int k = tid; k < nin; k += BT) xs[i * nin + k] = to_f32(xr[k]);
    for (int k = tid; k < H; k += BT) hs[i * H + k] = to_f32(hr[k]);
  }
  __syncthreads();
  // row group grp <-> gate q = grp / UN (i, f, g, o), unit u0 + grp % UN
  const int q = grp / UN, u = blockIdx.x * UN + grp % UN;
  const long row = (long)q * H + u;
  float a_ih[NBMAX], a_hh[NBMAX];
#pragma unroll
  for (int i = 0; i < NBMAX; ++i) { a_ih[i] = 0.f; a_hh[i] = 0.f; }
  row_dots<T>(static_cast<const T*>(a.w_ih) + row * nin, nin, xs, nin, nb, sub, a_ih);
  row_dots<T>(static_cast<const T*>(a.w_hh) + row * H, H, hs, H, nb, sub, a_hh);
  const float b = a.bias[row];
#pragma unroll
  for (int i = 0; i < NBMAX; ++i) {
    if (i < nb) {
      const float pre = rnd<T>(group_sum8(a_ih[i]) + b);          // (the chain stores the input product + bias, then adds the
      const float gate = rnd<T>(pre + group_sum8(a_hh[i]));       // recurrent product to it as a residual)
      if (sub == 0) gs[i * (4 * UN) + grp] = gate;
    }
  }
  __syncthreads();
  // cell update: thread <-> (hypothesis i, unit j)
  if (tid < nb * UN) {
    const int i = tid / UN, j = tid % UN, uu = blockIdx.x * UN + j;
    const float* g4 = gs + i * (4 * UN);
    const float ig = sigmoid_t<T>(g4[j]), fg = sigmoid_t<T>(g4[UN + j]), gg = tanh_t<T>(g4[2 * UN + j]), og = sigmoid_t<T>(g4[3 * UN + j]);
    const float cp = a.pc[cw[1][i] * (long)H + uu];
    const float cn = fg * cp + ig * gg;
    // (dst must not alias any hypothesis' src: other workgroups may still be staging h of that slot -- the search hands out
    // fresh destination slots every round, engine._rnnt_beam_search_graph)
    a.pc[cw[2][i] * (long)H + uu] = cn;
    static_cast<T*>(a.ph)[cw[2][i] * (long)H + uu] = from_f32<T>(og * tanh_t<T>(cn));
  }
}

struct BeamJointArgs {
  int nb, H, J, Tmax;
  const void* ph; const long long* dst;     // the top LSTM layer's new h: ph[dst[i]]
  const void* w_dec; const float* b_dec;    // [J][H], [J]
  const void* e_all; const long long* t;    // [Tmax][J] = w_enc . eouts + bias; frame index (device word)
  void* hj;                                 // [nb][J] joint input tanh(e_t + g)
};

template <typename T>
__global__ __launch_bounds__(BT) void rnnt_beam_joint_kernel(const BeamJointArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nb = a.nb, H = a.H, J = a.J, tid = threadIdx.x, grp = tid >> 3, sub = tid & 7;
  float* hs = reinterpret_cast<float*>(smem);   // [NBMAX][H]
  const T* ph = static_cast<const T*>(a.ph);
  __shared__ long long cw[NBMAX + 1];
  if (tid < nb) cw[tid] = a.dst[tid];
  if (tid == NBMAX) cw[NBMAX] = *a.t;
  __syncthreads();
  for (int i = 0; i < nb; ++i) {
    const T* hr = ph + cw[i] * (long)H;
    for (int k = tid; k < H; k += BT) hs[i * H + k] = to_f32(hr[k]);
  }
  __syncthreads();
  const int j = blockIdx.x * 32 + grp;          // 32 rows of w_dec per workgroup
  if (j >= J) return;
  float acc[NBMAX];
#pragma unroll
  for (int i = 0; i < NBMAX; ++i) acc[i] = 0.f;
  row_dots<T>(static_cast<const T*>(a.w_dec) + (long)j * H, H, hs, H, nb, sub, acc);
  long t = cw[NBMAX];
  t = t < 0 ? 0 : (t >= a.Tmax ? a.Tmax - 1 : t);
  const float e = to_f32(static_cast<const T*>(a.e_all)[t * J + j]);
  const float b = a.b_dec[j];
#pragma unroll
  for (int i = 0; i < NBMAX; ++i) {
    if (i < nb) {
      const float g = rnd<T>(group_sum8(acc[i]) + b);
      if (sub == 0) static_cast<T*>(a.hj)[(long)i * J + j] = from_f32<T>(tanh_t<T>(e + g));
    }
  }
}

// one WAVE per hypothesis (no workgroup barrier anywhere): lp = log_softmax(logits[i, :V]);  out[i] = { lp[blank], the k best of
// lp[1:] (descending, ties -> lowest index), their indices RELATIVE TO COLUMN 1 as floats } -- the record
// engine._rnnt_beam_search_graph reads.  The row lives in LDS (V floats); a lane scans every 64th entry.
template <typename T>
__global__ __launch_bounds__(64) void rnnt_beam_pick_kernel(int V, int k, int blank, const T* __restrict__ logits, long ldl,
                                                           float* __restrict__ out, long ldo) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* buf = reinterpret_cast<float*>(smem);   // [V]
  const long m = blockIdx.x;
  const int lane = threadIdx.x;
  const T* row = logits + m * ldl;
  float mx = -INFINITY;
  for (int v = lane; v < V; v += 64) { const float x = to_f32(row[v]); buf[v] = x; mx = fmaxf(mx, x); }
  mx = wave_max(mx);
  float se = 0.f;
  for (int v = lane; v < V; v += 64) se += expf(buf[v] - mx);
  se = wave_sum(se);
  const float lse = mx + logf(se);
  for (int v = lane; v < V; v += 64) buf[v] -= lse;
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {
    out[m * ldo] = buf[blank];
    buf[0] = -INFINITY;   // the candidates are columns 1 .. V - 1 (the chain's topk(lp[:, 1:]))
  }
  __builtin_amdgcn_wave_barrier();
  for (int j = 0; j < k; ++j) {
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int v = lane; v < V; v += 64) {
      const float c = buf[v];
      if (v >= 1 && (c > best || (c == best && v < bi))) { best = c; bi = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (bi == 0x7fffffff) bi = 1;
    if (lane == 0) {
      out[m * ldo + 1 + j] = best;
      out[m * ldo + 1 + k + j] = (float)(bi - 1);
      buf[bi] = -INFINITY;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// V <= 1024: the row in REGISTERS (16 values per lane, lane l holds columns l, l + 64, ...): no LDS, no dependent LDS round trips
// (the LDS form above takes 17.7 us for V = 1000: seven lane-strided passes of sixteen dependent reads each; this one ~3 us)
template <typename T>
__global__ __launch_bounds__(64) void rnnt_beam_pick16_kernel(int V, int k, int blank, const T* __restrict__ logits, long ldl,
                                                             float* __restrict__ out, long ldo) {
  const long m = blockIdx.x;
  const int lane = threadIdx.x;
  const T* row = logits + m * ldl;
  float x[16];
  float mx = -INFINITY;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int v = e * 64 + lane;
    x[e] = v < V ? to_f32(row[v]) : -INFINITY;
    mx = fmaxf(mx, x[e]);
  }
  mx = wave_max(mx);
  float se = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) se += (e * 64 + lane < V) ? expf(x[e] - mx) : 0.f;
  se = wave_sum(se);
  const float lse = mx + logf(se);
  float lpb = -INFINITY;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    x[e] -= lse;
    if (e * 64 + lane == blank) lpb = x[e];
    if (e * 64 + lane == 0) x[e] = -INFINITY;   // the candidates are columns 1 .. V - 1 (the chain's topk(lp[:, 1:]))
  }
  lpb = wave_max(lpb);
  if (lane == 0) out[m * ldo] = lpb;
  for (int j = 0; j < k; ++j) {
    float best = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
    for (int e = 0; e < 16; ++e) {   // ascending columns: a strict > keeps the lowest index among equals
      if (x[e] > best) { best = x[e]; bi = e * 64 + lane; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (bi == 0x7fffffff) bi = 1;
    if (lane == 0) {
      out[m * ldo + 1 + j] = best;
      out[m * ldo + 1 + k + j] = (float)(bi - 1);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e)
      if (e * 64 + lane == bi) x[e] = -INFINITY;
  }
}

// ---- bf16: the same two kernels on the matrix cores --------------------------------------------------------------------------------
// A wave computes D[16 weight rows x 16 hypotheses] = W[16 rows][K] . X^T with v_mfma_f32_16x16x32_bf16: the A fragments (16 bytes of
// one weight row per lane) come straight from global memory -- every 16-byte piece of the 4 MB of LSTM weights is read by exactly one
// lane of the launch, all of a wave's loads are in flight together -- the B fragments (16 bytes of one hypothesis' input row) from an
// LDS image [16][K + 8] (bf16; rows of hypotheses >= nb are zero).  The VALU form above spent its time re-reading the inputs from LDS
// for every weight row (13 us per layer at H = 512; this form: weight-streaming latency).
typedef __attribute__((ext_vector_type(4))) float f32x4_;
__device__ __forceinline__ f32x4_ rows16_dot(const bf16* __restrict__ wrow0, long ldw, int K, const bf16* __restrict__ xs, int ldxs,
                                             int lane) {
  // wrow0: first of the wave's 16 rows; lane <-> (row lane & 15, k piece 8 * (lane >> 4)) of each 32-wide k step
  const bf16* wp = wrow0 + (long)(lane & 15) * ldw + 8 * (lane >> 4);
  const bf16* xp = xs + (lane & 15) * ldxs + 8 * (lane >> 4);
  f32x4_ acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int k = 0; k < K; k += 32) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(wp + k);
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(xp + k);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  }
  return acc;   // acc[r] = D[row 4 * (lane >> 4) + r][hypothesis lane & 15]
}

// grid = H / 16 workgroups of 4 waves: wave q = gate q (i, f, g, o) of the workgroup's 16 hidden units
__global__ __launch_bounds__(BT) void rnnt_beam_lstm_mfma_kernel(const BeamLstmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nb = a.nb, nin = a.nin, H = a.H, tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
  const int ldx = nin + 8, ldh = H + 8;
  bf16* xs = reinterpret_cast<bf16*>(smem);              // [16][nin + 8]
  bf16* hs = xs + NBMAX * ldx;                           // [16][H + 8]
  float* gs = reinterpret_cast<float*>(hs + NBMAX * ldh);   // [16 hypotheses][4 gates][16 units]
  const bf16* xtab = static_cast<const bf16*>(a.xtab);
  const bf16* ph = static_cast<const bf16*>(a.ph);
  __shared__ long long cw[3][NBMAX];   // control words (possibly in pinned host memory): read once
  if (tid < nb) { cw[0][tid] = a.xidx[tid]; cw[1][tid] = a.src[tid]; cw[2][tid] = a.dst[tid]; }
  if (blockIdx.x == 0 && a.copy_dst && tid >= 64 && tid - 64 < a.copy_n) a.copy_dst[tid - 64] = a.copy_src[tid - 64];
  __syncthreads();
  // stage the inputs: 16-byte pieces, zero rows for hypotheses >= nb
  for (int p = tid; p < NBMAX * (nin / 8); p += BT) {
    const int i = p / (nin / 8), c = (p % (nin / 8)) * 8;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
    if (i < nb) v = *reinterpret_cast<const bf16x8*>(xtab + cw[0][i] * a.ldx + c);
    *reinterpret_cast<bf16x8*>(xs + i * ldx + c) = v;
  }
  for (int p = tid; p < NBMAX * (H / 8); p += BT) {
    const int i = p / (H / 8), c = (p % (H / 8)) * 8;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
    if (i < nb) v = *reinterpret_cast<const bf16x8*>(ph + cw[1][i] * (long)H + c);
    *reinterpret_cast<bf16x8*>(hs + i * ldh + c) = v;
  }
  __syncthreads();
  const int u0 = blockIdx.x * 16;
  const long row0 = (long)q * H + u0;
  const f32x4_ d_ih = rows16_dot(static_cast<const bf16*>(a.w_ih) + row0 * nin, nin, nin, xs, ldx, lane);
  const f32x4_ d_hh = rows16_dot(static_cast<const bf16*>(a.w_hh) + row0 * H, H, H, hs, ldh, lane);
  {
    const int i = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 4 * (lane >> 4) + r;   // unit inside the workgroup
      const float pre = rnd<bf16>(d_ih[r] + a.bias[row0 + j]);
      gs[(i * 4 + q) * 16 + j] = rnd<bf16>(pre + d_hh[r]);
    }
  }
  __syncthreads();
  if (tid < nb * 16) {
    const int i = tid / 16, j = tid % 16, uu = u0 + j;
    const float* g4 = gs + i * 64;
    const float ig = sigmoid_t<bf16>(g4[j]), fg = sigmoid_t<bf16>(g4[16 + j]), gg = tanh_t<bf16>(g4[32 + j]), og = sigmoid_t<bf16>(g4[48 + j]);
    const float cn = fg * a.pc[cw[1][i] * (long)H + uu] + ig * gg;
    a.pc[cw[2][i] * (long)H + uu] = cn;
    static_cast<bf16*>(a.ph)[cw[2][i] * (long)H + uu] = (bf16)(og * tanh_t<bf16>(cn));
  }
}

// grid = ceil(J / 64) workgroups of 4 waves, 16 rows of w_dec per wave
__global__ __launch_bounds__(BT) void rnnt_beam_joint_mfma_kernel(const BeamJointArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nb = a.nb, H = a.H, J = a.J, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ldh = H + 8;
  bf16* hs = reinterpret_cast<bf16*>(smem);   // [16][H + 8]
  const bf16* ph = static_cast<const bf16*>(a.ph);
  __shared__ long long cw[NBMAX + 1];
  if (tid < nb) cw[tid] = a.dst[tid];
  if (tid == NBMAX) cw[NBMAX] = *a.t;
  __syncthreads();
  for (int p = tid; p < NBMAX * (H / 8); p += BT) {
    const int i = p / (H / 8), c = (p % (H / 8)) * 8;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
    if (i < nb) v = *reinterpret_cast<const bf16x8*>(ph + cw[i] * (long)H + c);
    *reinterpret_cast<bf16x8*>(hs + i * ldh + c) = v;
  }
  __syncthreads();
  const int j0 = blockIdx.x * 64 + 16 * w;
  if (j0 >= J) return;                        // (J % 16 == 0: a wave's 16 rows are all inside or all outside)
  const f32x4_ d = rows16_dot(static_cast<const bf16*>(a.w_dec) + (long)j0 * H, H, H, hs, ldh, lane);
  long t = cw[NBMAX];
  t = t < 0 ? 0 : (t >= a.Tmax ? a.Tmax - 1 : t);
  const int i = lane & 15;
  if (i >= nb) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = j0 + 4 * (lane >> 4) + r;
    const float g = rnd<bf16>(d[r] + a.b_dec[j]);
    const float e = (float)static_cast<const bf16*>(a.e_all)[t * J + j];
    static_cast<bf16*>(a.hj)[(long)i * J + j] = (bf16)tanh_t<bf16>(e + g);
  }
}

}  // namespace

void emo_rnnt_set_beam_mfma(int v) { g_beam_mfma = v ? 1 : 0; }

extern "C" int emoasr_rnnt_beam_lstm(int dtype, int nb, int nin, int H, const void* xtab, long ldx, const long long* xidx,
                                     const void* w_ih, const void* w_hh, const float* bias, void* ph, float* pc,
                                     const long long* src, const long long* dst, const long long* copy_src, long long* copy_dst,
                                     int copy_n, void* stream) {
  EMO_CHECK(copy_n >= 0 && copy_n <= BT - 64, "rnnt_beam_lstm: copy_n=%d too long", copy_n);
  EMO_CHECK(nb >= 1 && nb <= NBMAX, "rnnt_beam_lstm: nb=%d outside 1..%d", nb, NBMAX);
  const int vec = dtype == EMO_BF16 ? 8 : 4;
  EMO_CHECK(H % UN == 0 && nin % vec == 0 && H % vec == 0 && ldx % vec == 0, "rnnt_beam_lstm: H=%d nin=%d unsupported", H, nin);
  const size_t smem = ((size_t)NBMAX * (nin + H) + NBMAX * 4 * UN) * sizeof(float);
  EMO_CHECK(smem <= 150 * 1024, "rnnt_beam_lstm: nin + H = %d too wide for the LDS plan", nin + H);
  BeamLstmArgs a{nb, nin, H, xtab, ldx, xidx, w_ih, w_hh, bias, ph, pc, src, dst, copy_src, copy_dst, copy_n};
  if (dtype == EMO_BF16 && g_beam_mfma && H % 32 == 0 && nin % 32 == 0) {
    const size_t sm = (size_t)NBMAX * (nin + 8 + H + 8) * 2 + NBMAX * 64 * sizeof(float);
    EMO_CHECK(sm <= 64 * 1024, "rnnt_beam_lstm: nin + H = %d too wide for the LDS plan", nin + H);
    rnnt_beam_lstm_mfma_kernel<<<H / 16, BT, sm, (hipStream_t)stream>>>(a);
    EMO_LAUNCH_CHECK();
    return 0;
  }
  EMO_DISPATCH(dtype, {
    static size_t set_bytes = 0;
    if (smem > 64 * 1024 && smem > set_bytes) {
      EMO_CHECK(hipFuncSetAttribute((const void*)rnnt_beam_lstm_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) == hipSuccess,
                "rnnt_beam_lstm: hipFuncSetAttribute(%zu) failed", smem);
      set_bytes = smem;
    }
    rnnt_beam_lstm_kernel<T><<<H / UN, BT, smem, (hipStream_t)stream>>>(a);
  });
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_rnnt_beam_joint(int dtype, int nb, int H, int J, int Tmax, const void* ph, const long long* dst,
                                      const void* w_dec, const float* b_dec, const void* e_all, const long long* t, void* hj,
                                      void* stream) {
  EMO_CHECK(nb >= 1 && nb <= NBMAX && Tmax >= 1, "rnnt_beam_joint: nb=%d outside 1..%d", nb, NBMAX);
  const int vec = dtype == EMO_BF16 ? 8 : 4;
  EMO_CHECK(H % vec == 0, "rnnt_beam_joint: H=%d unsupported", H);
  const size_t smem = (size_t)NBMAX * H * sizeof(float);
  EMO_CHECK(smem <= 64 * 1024, "rnnt_beam_joint: H=%d too wide", H);
  BeamJointArgs a{nb, H, J, Tmax, ph, dst, w_dec, b_dec, e_all, t, hj};
  if (dtype == EMO_BF16 && g_beam_mfma && H % 32 == 0 && J % 16 == 0 && (size_t)NBMAX * (H + 8) * 2 <= 64 * 1024) {
    rnnt_beam_joint_mfma_kernel<<<(J + 63) / 64, BT, (size_t)NBMAX * (H + 8) * 2, (hipStream_t)stream>>>(a);
    EMO_LAUNCH_CHECK();
    return 0;
  }
  EMO_DISPATCH(dtype, (rnnt_beam_joint_kernel<T><<<(J + 31) / 32, BT, smem, (hipStream_t)stream>>>(a)));
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_rnnt_beam_pick(int dtype, int nb, int V, int k, int blank, const void* logits, long ldl, float* out, long ldo,
                                     void* stream) {
  EMO_CHECK(nb >= 1 && k >= 1 && k < V && blank >= 0 && blank < V && ldo >= 1 + 2 * k, "rnnt_beam_pick: bad arguments");
  EMO_CHECK((size_t)V * 4 <= 60 * 1024, "rnnt_beam_pick: V=%d too large for an LDS row", V);
  if (V <= 1024) {
    EMO_DISPATCH(dtype, (rnnt_beam_pick16_kernel<T><<<nb, 64, 0, (hipStream_t)stream>>>(V, k, blank, (const T*)logits, ldl, out, ldo)));
    EMO_LAUNCH_CHECK();
    return 0;
  }
  EMO_DISPATCH(dtype, (rnnt_beam_pick_kernel<T><<<nb, 64, (size_t)V * 4, (hipStream_t)stream>>>(V, k, blank, (const T*)logits, ldl,
                                                                                              out, ldo)));
  EMO_LAUNCH_CHECK();
  return 0;
}
