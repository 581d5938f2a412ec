// Common device helpers for the emoasr_amd HIP kernels (gfx950 / CDNA4 only).
//
// Conventions
//   * wave = 64 lanes; blocks are multiples of 64 threads.
//   * T is the "compute" element type: float (parity mode, exact-f32 MFMA) or
//     __bf16 (throughput mode, bf16 MFMA with f32 accumulation).
//   * statistics, reductions, softmax / LSE, lattices and gradients of
//     parameters are always f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define EMO_WAVE 64

// ---------------------------------------------------------------------------
// scalar conversion
// ---------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16>(bf16 v) { return (float)v; }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// ---------------------------------------------------------------------------
// 16-byte vectors of T: 4 floats or 8 bf16.
// ---------------------------------------------------------------------------
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  f32x4 v;
  __device__ __forceinline__ float get(int i) const { return v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
  __device__ __forceinline__ void zero() { v = f32x4{0.f, 0.f, 0.f, 0.f}; }
};
template <> struct Vec16<bf16> {
  static constexpr int N = 8;
  bf16x8 v;
  __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16)x; }
  __device__ __forceinline__ void zero() {
    for (int i = 0; i < 8; ++i) v[i] = (bf16)0.f;
  }
};

// f32s: "split" element type of the attention kernels' f32x3 mode (csrc/mma.h Mma<f32s>): plain f32 in HBM and in the kernels'
// f32 images; only the MFMA operand tiles in LDS hold (hi, lo) bf16 pairs packed into the element's 32 bits
struct f32s { float x; };
template <> __device__ __forceinline__ float to_f32<f32s>(f32s v) { return v.x; }
template <> __device__ __forceinline__ f32s from_f32<f32s>(float v) { return f32s{v}; }
template <> struct Vec16<f32s> {
  static constexpr int N = 4;
  f32x4 v;
  __device__ __forceinline__ float get(int i) const { return v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
  __device__ __forceinline__ void zero() { v = f32x4{0.f, 0.f, 0.f, 0.f}; }
};

template <typename T>
__device__ __forceinline__ Vec16<T> load16(const T* p) {
  Vec16<T> r;
  r.v = *reinterpret_cast<const decltype(r.v)*>(p);
  return r;
}
template <typename T>
__device__ __forceinline__ void store16(T* p, const Vec16<T>& r) {
  *reinterpret_cast<decltype(r.v)*>(p) = r.v;
}

// ---------------------------------------------------------------------------
// Bounds-checked buffer loads.  An offset of EMO_OOB (or anything past the descriptor's size)
// returns zeros, so a row / time / k guard is one v_cndmask on the offset: no exec-mask branch and
// no s_waitcnt vmcnt(0) per guarded load (what `cond ? *p : 0` compiles to), i.e. a run of guarded
// loads stays in flight together.  Offsets are bytes and 32-bit: operands must be < 4 GiB.
// ---------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define EMO_OOB 0xFFFFFFFFu
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0xFFFFFFFF, 0x00020000);
}
// descriptor of exactly `bytes` bytes: every offset past the operand's last valid byte reads zeros BY THE DESCRIPTOR -- row guards
// need no per-load compare / select at all (negative row indices wrap to huge unsigned offsets: out of range as well)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_n(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
template <typename T>
__device__ __forceinline__ Vec16<T> buf_load16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
  Vec16<T> o;
  o.v = __builtin_bit_cast(decltype(o.v), v);
  return o;
}
// one element of T (2 or 4 bytes) as f32
template <typename T>
__device__ __forceinline__ float buf_load_f32(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  if constexpr (sizeof(T) == 2) {
    const unsigned short u = __builtin_amdgcn_raw_buffer_load_b16(r, byte_off, 0, 0);
    return __uint_as_float((unsigned)u << 16);
  } else {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
  }
}

// one element of T from an f32 value; an out-of-range offset (EMO_OOB) drops the store
template <typename T>
__device__ __forceinline__ void buf_store_elem(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float v) {
  if constexpr (sizeof(T) == 2) {
    const bf16 h = (bf16)v;
    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, h), r, byte_off, 0, 0);
  } else {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, byte_off, 0, 0);
  }
}
// 8 consecutive elements of T as f32 (one 16-byte load for bf16, two for f32); zeros when !ok
template <typename T>
__device__ __forceinline__ void buf_load8(__amdgpu_buffer_rsrc_t r, long elem_off, bool ok, float (&o)[8]) {
  const unsigned off = ok ? (unsigned)(elem_off * (long)sizeof(T)) : EMO_OOB;
  const Vec16<T> v = buf_load16<T>(r, off);
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = v.get(j);
  } else {
    const Vec16<T> w = buf_load16<T>(r, ok ? off + 16u : EMO_OOB);
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = v.get(j); o[4 + j] = w.get(j); }
  }
}
template <typename T>
__device__ __forceinline__ void load8(const T* p, float (&o)[8]) {
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
__device__ __forceinline__ void store8(T* p, const float (&o)[8]) {
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

// ---------------------------------------------------------------------------
// wave / block reductions (f32)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum of one float per thread; `red` is LDS scratch of >= 16 floats.
// All threads get the result.  blockDim.x must be a multiple of 64.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
  for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
  return t;
}

// One fused multiply-add per tap in the depthwise stencils (the build has -ffp-contract=off: a * b + c stays two instructions
// and two roundings elsewhere, which the bit-identity tests between fused and unfused kernels rely on).  The convolution kernels
// are VALU-bound (31 taps per output, forward, data gradient and weight gradient): v_fma_f32 / v_pk_fma_f32 halve their inner
// loops.  Every stencil -- convmodule.hip and convfused.hip -- goes through these two, so they stay bit-identical to each other.
typedef __attribute__((ext_vector_type(2))) float emo_f2;
#ifdef EMO_CONV_NO_FMA   // A/B builds only
__device__ __forceinline__ float emo_mac(float a, float b, float c) { return c + a * b; }
__device__ __forceinline__ emo_f2 emo_mac2(emo_f2 a, emo_f2 b, emo_f2 c) { return c + a * b; }
#else
__device__ __forceinline__ float emo_mac(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ emo_f2 emo_mac2(emo_f2 a, emo_f2 b, emo_f2 c) { return __builtin_elementwise_fma(a, b, c); }
#endif

// ---------------------------------------------------------------------------
// activations
// ---------------------------------------------------------------------------
// v_rcp_f32 (1 ulp) instead of the IEEE division sequence (div_scale x2, rcp, 6 fma / mul, div_fmas, div_fixup: 11 VALU operations
// per value -- the Swish epilogue of the first feed-forward product and the GLU / Swish staging of the convolution kernels take one
// sigmoid per element).  The hardware forms serve the bf16 instantiations ONLY: the f32 engine is the parity mode and keeps the
// IEEE division and the library tanhf (sigmoid_t / tanh_t / swish_t / dswish_t below pick by the kernel's element type).
#ifdef EMO_SIGMOID_IEEE_DIV   // A/B builds only (python -m emoasr_amd.build --variant div -DEMO_SIGMOID_IEEE_DIV)
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
#else
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
#endif
__device__ __forceinline__ float sigmoid_exact(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float swishf_(float x) { return x * sigmoidf_(x); }
// tanh on the hardware exp / rcp units: 1 - 2 / (exp(2x) + 1) (inf for large x -> 1, 0 for very negative x -> -1).  ABSOLUTE error
// ~1e-7 (the subtraction cancels near 0: the relative error there is ~1e-7 / |x|), which is below bf16's rounding and not good
// enough for a parity claim: bf16 kernels only.  tanhf's library expansion is ~30 instructions with a division: the LSTM cells
// take two per (sequence, unit) and position on the recurrence's critical path.
__device__ __forceinline__ float tanh_fast(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f); }
__device__ __forceinline__ float dswishf_(float x) {
  const float s = sigmoidf_(x);
  return s * (1.f + x * (1.f - s));
}
// element-type dispatch: float -> exact forms, bf16 -> hardware forms
template <typename T> __device__ __forceinline__ float sigmoid_t(float x) {
  if constexpr (sizeof(T) == 4) return sigmoid_exact(x); else return sigmoidf_(x);
}
template <typename T> __device__ __forceinline__ float tanh_t(float x) {
  if constexpr (sizeof(T) == 4) return tanhf(x); else return tanh_fast(x);
}
template <typename T> __device__ __forceinline__ float swish_t(float x) { return x * sigmoid_t<T>(x); }
template <typename T> __device__ __forceinline__ float dswish_t(float x) {
  const float s = sigmoid_t<T>(x);
  return s * (1.f + x * (1.f - s));
}

enum { EMO_ACT_NONE = 0, EMO_ACT_RELU = 1, EMO_ACT_SWISH = 2, EMO_ACT_GELU = 3, EMO_ACT_TANH = 4,
       EMO_DACT_TANH_OUT = 5,
       EMO_DACT_MUL = 6 };        // epilogue.dact: the saved tensor IS the factor (see EMO_ACT_SAVE_DACT)
// epilogue.act | EMO_ACT_SAVE_DACT: `pre_out` receives act'(pre) * dropout_scale -- the factor the data gradient multiplies by --
// instead of the pre-activation.  The backward epilogue (dact = EMO_DACT_MUL, drop_p = 0) is then one multiply per element: no
// sigmoid and no second hashing of the dropout mask (16 of its ~25 VALU operations per element).
constexpr int EMO_ACT_SAVE_DACT = 0x100;

__device__ __forceinline__ float geluf_(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgeluf_(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

template <typename T = bf16>
__device__ __forceinline__ float apply_act(int act, float x) {
  if (act == EMO_ACT_RELU) return fmaxf(x, 0.f);
  if (act == EMO_ACT_SWISH) return swish_t<T>(x);
  if (act == EMO_ACT_GELU) return geluf_(x);
  if (act == EMO_ACT_TANH) return tanhf(x);
  return x;
}
template <typename T = bf16>
__device__ __forceinline__ float apply_dact(int act, float pre) {
  if (act == EMO_ACT_RELU) return pre > 0.f ? 1.f : 0.f;
  if (act == EMO_ACT_SWISH) return dswish_t<T>(pre);
  if (act == EMO_ACT_GELU) return dgeluf_(pre);
  if (act == EMO_ACT_TANH) { const float t = tanhf(pre); return 1.f - t * t; }
  if (act == EMO_DACT_TANH_OUT) return 1.f - pre * pre;  // `pre` holds tanh's OUTPUT here
  if (act == EMO_DACT_MUL) return pre;
  return 1.f;
}

// Vector forms: ONE dispatch on the (wave-uniform) activation code per group of N values.  Calling
// apply_act / apply_dact per element inside an unrolled loop makes the compiler emit the whole
// compare-and-branch ladder (with the erf / tanh / exp expansions behind it) once per element: the GEMM
// epilogue was ~12 000 instructions and ~900 scalar branches long, most of a K = 256 tile's time.
template <int N, typename T = bf16>
__device__ __forceinline__ void act_vec(int act, float (&v)[N]) {
  switch (act) {
    case EMO_ACT_RELU:
#pragma unroll
      for (int e = 0; e < N; ++e) v[e] = fmaxf(v[e], 0.f);
      break;
    case EMO_ACT_SWISH:
#pragma unroll
      for (int e = 0; e < N; ++e) v[e] = swish_t<T>(v[e]);
      break;
    case EMO_ACT_GELU:
#pragma unroll 1
      for (int e = 0; e < N; ++e) v[e] = geluf_(v[e]);
      break;
    case EMO_ACT_TANH:
#pragma unroll 1
      for (int e = 0; e < N; ++e) v[e] = tanhf(v[e]);
      break;
    default: break;
  }
}
// v[e] *= act'(pre[e])   (EMO_DACT_TANH_OUT: pre holds tanh's output)
template <int N, typename T = bf16>
__device__ __forceinline__ void dact_vec(int act, const float (&pre)[N], float (&v)[N]) {
  switch (act) {
    case EMO_ACT_RELU:
#pragma unroll
      for (int e = 0; e < N; ++e) v[e] = pre[e] > 0.f ? v[e] : 0.f;
      break;
    case EMO_ACT_SWISH:
#pragma unroll
      for (int e = 0; e < N; ++e) v[e] *= dswish_t<T>(pre[e]);
      break;
    case EMO_DACT_MUL:
#pragma unroll
      for (int e = 0; e < N; ++e) v[e] *= pre[e];
      break;
    case EMO_ACT_GELU:
#pragma unroll 1
      for (int e = 0; e < N; ++e) v[e] *= dgeluf_(pre[e]);
      break;
    case EMO_ACT_TANH:
#pragma unroll 1
      for (int e = 0; e < N; ++e) { const float t = tanhf(pre[e]); v[e] *= 1.f - t * t; }
      break;
    case EMO_DACT_TANH_OUT:
#pragma unroll
      for (int e = 0; e < N; ++e) v[e] *= 1.f - pre[e] * pre[e];
      break;
    default: break;
  }
}

// ---------------------------------------------------------------------------
// Counter-based dropout RNG.  keep(seed, idx) is a pure function so backward
// kernels regenerate the forward mask instead of storing it.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t hash_u32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU;
  x ^= x >> 15; x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
// keep decision of element idx (p > 0).  Elements are hashed in PAIRS: four xorshift / 24-bit-multiply rounds over
// (idx >> 1, seed) give the 24-bit uniform of the even element, one more round gives the odd element's -- a kernel that walks
// consecutive elements (the attention tiles: 16 per lane and step) pays 9 integer operations per decision instead of 17, and
// the compare is against an integer threshold (no conversion).  The masks are regenerated in every backward kernel, so this
// sits on the VALU critical path of the attention kernels (1.4 k of 9 k cycles per step before pairing, in-kernel stamps):
// v_mul_u32_u24 issues at full rate where the 32-bit v_mul_lo_u32 of a classic avalanche hash is quarter rate.
// Statistics (tools/hash_check.py, numpy restatement): chi-square over 64 buckets of 4M consecutive indices 57-96 for 63
// degrees of freedom for both elements of a pair; adjacent-index / row-stride / seed+1 / within-pair correlations of the keep
// mask < 3e-3.
__device__ __forceinline__ uint32_t dropout_hash(uint64_t seed, uint64_t pair) {
  const uint32_t lo = (uint32_t)pair, hi = (uint32_t)(pair >> 32);
  uint32_t x = lo ^ (uint32_t)seed ^ __umul24(hi, 0x85EBCBu);
  x ^= x >> 16;
  x = __umul24(x, 0x9E3779u) ^ (uint32_t)(seed >> 32);
  x ^= x >> 13;
  x = __umul24(x, 0xC2B2AFu);
  x ^= x >> 15;
  x = __umul24(x, 0x7FEB35u);
  x ^= x >> 12;
  return x;
}
__device__ __forceinline__ uint32_t dropout_second(uint32_t x) {
  const uint32_t y = __umul24(x ^ (x >> 11), 0x9E3779u);
  return y ^ (y >> 14);
}
// keep <=> (24-bit uniform) >= thr, i.e. u24 * 2^-24 >= p exactly
__device__ __forceinline__ uint32_t dropout_thr(float p) { return (uint32_t)ceilf(p * 16777216.f); }
// the two elements 2 * pair and 2 * pair + 1
__device__ __forceinline__ void dropout_keep2(uint64_t seed, uint64_t pair, uint32_t thr, bool& k0, bool& k1) {
  const uint32_t x = dropout_hash(seed, pair);
  k0 = (x & 0xFFFFFFu) >= thr;
  k1 = (dropout_second(x) & 0xFFFFFFu) >= thr;
}
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint64_t idx, float p) {
  const uint32_t x = dropout_hash(seed, idx >> 1);
  const uint32_t u = ((idx & 1) ? dropout_second(x) : x) & 0xFFFFFFu;
  return u >= dropout_thr(p);
}
// returns 0.f (dropped) or 1/(1-p) (kept).  p == 0 -> always 1.
__device__ __forceinline__ float dropout_scale(uint64_t seed, uint64_t idx, float p) {
  if (p <= 0.f) return 1.f;
  return dropout_keep(seed, idx, p) ? 1.f / (1.f - p) : 0.f;
}

// v[e] *= dropout_scale(seed, idx0 + e, p) for 8 CONSECUTIVE elements starting at an EVEN index: four hashes instead of eight
// (a pair shares one hash by construction; called per element the pair could not be merged by the compiler, which cannot see that
// row * N + col is even), the threshold and 1 / (1 - p) formed once.  Same mask, bit for bit, as the per-element form.
__device__ __forceinline__ void dropout_apply8(uint64_t seed, uint64_t idx0, float p, float (&v)[8]) {
  if (p <= 0.f) return;
  const uint32_t thr = dropout_thr(p);
  const float inv = 1.f / (1.f - p);
  const uint64_t pair0 = idx0 >> 1;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t x = dropout_hash(seed, pair0 + (uint64_t)k);
    v[2 * k] *= (x & 0xFFFFFFu) >= thr ? inv : 0.f;
    v[2 * k + 1] *= (dropout_second(x) & 0xFFFFFFu) >= thr ? inv : 0.f;
  }
}

// m[e] = dropout_scale(seed, idx0 + e, p) for 8 consecutive elements from an even index (the multipliers of dropout_apply8)
__device__ __forceinline__ void dropout_mult8(uint64_t seed, uint64_t idx0, float p, float (&m)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) m[e] = 1.f;
  if (p <= 0.f) return;
  const uint32_t thr = dropout_thr(p);
  const float inv = 1.f / (1.f - p);
  const uint64_t pair0 = idx0 >> 1;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t x = dropout_hash(seed, pair0 + (uint64_t)k);
    m[2 * k] = (x & 0xFFFFFFu) >= thr ? inv : 0.f;
    m[2 * k + 1] = (dropout_second(x) & 0xFFFFFFu) >= thr ? inv : 0.f;
  }
}
// v <- act(v), d <- act'(v) in one go (Swish: one sigmoid for both)
template <int N, typename T = bf16>
__device__ __forceinline__ void act_dact_vec(int act, float (&v)[N], float (&d)[N]) {
  if (act == EMO_ACT_SWISH) {
#pragma unroll
    for (int e = 0; e < N; ++e) {
      const float s = sigmoid_t<T>(v[e]), a = v[e] * s;
      d[e] = s + a * (1.f - s);
      v[e] = a;
    }
    return;
  }
#pragma unroll
  for (int e = 0; e < N; ++e) d[e] = 1.f;
  dact_vec<N, T>(act, v, d);
  act_vec<N, T>(act, v);
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. waits for every outstanding global
// store (the lattice kernels write one value per frame / diagonal) and load (their operands are prefetched chunks ahead) --
// neither is part of the exchange the barrier protects.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// ---------------------------------------------------------------------------
// log-space helpers for the CTC / RNN-T lattices
// ---------------------------------------------------------------------------
#define EMO_NEG_INF (-INFINITY)
__device__ __forceinline__ float log_add(float a, float b) {
  if (a == EMO_NEG_INF) return b;
  if (b == EMO_NEG_INF) return a;
  const float m = fmaxf(a, b);
  return m + log1pf(expf(-fabsf(a - b)));
}
__device__ __forceinline__ float log_add3(float a, float b, float c) {
  const float m = fmaxf(a, fmaxf(b, c));
  if (m == EMO_NEG_INF) return EMO_NEG_INF;
  return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}

// ---------------------------------------------------------------------------
// host-side launch helpers
// ---------------------------------------------------------------------------
void emo_set_error(const char* fmt, ...);
// Small device scratch areas that kernels of ONE stream hand from launch to launch (arrival tickets, partial sums, barrier
// counters): one zero-initialised area per (device, stream, slot), so that launches on different streams -- the attention
// backward's side stream, the transducer lattice under the CTC branch, two engines on two streams -- never share one.  `host` is a
// few words of host-side state that belong to the same area (the cooperative LSTM's expected barrier values).  Allocation happens
// on first use and needs an eager (non-capturing) call; a stream that is being captured may only use an area it has used before.
struct EmoScratch { void* dev; size_t bytes; unsigned host[32]; };
enum { EMO_SCRATCH_BN_TICKETS = 0, EMO_SCRATCH_SQNORM = 1, EMO_SCRATCH_LSTM = 2 };
EmoScratch* emo_stream_scratch(int slot, void* stream, size_t bytes);
#define EMO_CHECK(cond, ...)            \
  do {                                  \
    if (!(cond)) {                      \
      emo_set_error(__VA_ARGS__);       \
      return 1;                         \
    }                                   \
  } while (0)
#define EMO_LAUNCH_CHECK()                                              \
  do {                                                                  \
    hipError_t e__ = hipGetLastError();                                 \
    if (e__ != hipSuccess) {                                            \
      emo_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,      \
                    hipGetErrorString(e__));                            \
      return 2;                                                         \
    }                                                                   \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// HIP-event timers around selected kernels (api.hip; read with emoasr_timer_read, enabled by option "timers")
enum { EMO_TIMER_ATTN_BWD_MAIN = 0, EMO_TIMER_ATTN_BWD_DPOS = 1, EMO_TIMER_ATTN_FWD = 2, EMO_TIMER_TN_GROUPED = 3,
       EMO_TIMER_GEMM_NT_NN = 4, EMO_TIMER_GEMM_TN = 5, EMO_TIMER_LAYERNORM = 6, EMO_TIMER_CONV_MODULE = 7,
       EMO_TIMER_COUNT = 8 };
// option "timers": 0 off, 1 every family, else a bit mask (1 << (id + 1)) of the families to record
void emo_timer_begin(int id, hipStream_t s, double flops = 0.0, double bytes = 0.0);
void emo_timer_end(int id, hipStream_t s);
// records the launches enqueued on `s` during its lifetime (with their algorithmic work) under family `id`
struct EmoTimerScope {
  int id; hipStream_t s;
  EmoTimerScope(int id_, hipStream_t s_, double flops = 0.0, double bytes = 0.0) : id(id_), s(s_) { emo_timer_begin(id, s, flops, bytes); }
  ~EmoTimerScope() { emo_timer_end(id, s); }
};

// EMO_F32X3 (round 6): f32 storage, statistics and epilogues exactly as EMO_F32, every matrix product as three bf16 MFMAs over
// (hi, lo) operand pairs (csrc/gemm.hip SplitCfg, csrc/mma.h Mma<f32s>).  The mode travels in the dtype argument of every call --
// there is no library state behind it (rounds 4-5: a process-wide option) -- and every entry point that is not a product treats it
// as EMO_F32 (EMO_DISPATCH; element size 4).
enum { EMO_F32 = 0, EMO_BF16 = 1, EMO_F32X3 = 2 };
__host__ __device__ inline bool emo_is_f32(int dtype) { return dtype == EMO_F32 || dtype == EMO_F32X3; }
#define EMO_DISPATCH(dtype, ...)                                        \
  do {                                                                  \
    if ((dtype) == EMO_F32 || (dtype) == EMO_F32X3) { typedef float T; __VA_ARGS__; } \
    else if ((dtype) == EMO_BF16) { typedef bf16 T; __VA_ARGS__; }      \
    else { emo_set_error("bad dtype %d", (int)(dtype)); return 1; }     \
  } while (0)

// Stacked micro-batches for the per-utterance kernels of the convolution module (include/emoasr_hip.h: emoasr_segments_t):
// segment s = utterances b0[s] .. b0[s+1]-1, each padded to T[s] frames, rows row[s] .. row[s+1]-1 of the stacked [M, C] arrays;
// part[s] / sums[s] = first float / first partial row of its area in the BatchNorm forward / backward partial-sum tables.
// n <= 1: one dense batch (the kernels ignore the table).  Passed BY VALUE as a kernel argument (constant memory).
struct RowSegs {
  int n;
  int b0[9], T[8];
  long row[9], part[9], sums[9];
};
__device__ __forceinline__ int rowsegs_of_utt(const RowSegs& sg, int b) {
  int s = 0;
  for (int k = 1; k < 8; ++k) s += (k < sg.n && b >= sg.b0[k]) ? 1 : 0;
  return s;
}
__device__ __forceinline__ int rowsegs_of_row(const RowSegs& sg, long row) {
  int s = 0;
  for (int k = 1; k < 8; ++k) s += (k < sg.n && row >= sg.row[k]) ? 1 : 0;
  return s;
}
