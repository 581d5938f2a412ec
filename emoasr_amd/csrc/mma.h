// MFMA tile abstraction: one 32x32 output tile per wave-instruction.
//
//   bf16 : v_mfma_f32_32x32x16_bf16   (KSTEP = 16, 8 bf16 per lane per operand)
//   f32  : v_mfma_f32_32x32x2_f32     (KSTEP = 2, one f32 per lane per operand;
//                                       bit-exact k-ordered fmaf chain)
//
// Operand lane maps (gfx950):
//   bf16  A[row = lane&31][k = 8*(lane>>5) + j], j = 0..7   (B likewise, col = lane&31)
//   f32   A[row = lane&31][k = lane>>5]
// Accumulator (both): col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
//
// LDS tiles come in two layouts:
//   "kc" (k-contiguous) : element (row, k) at base[row*ld + k]
//   "km" (k-major)      : element (k, row) at base[k*ld + row]
// km/bf16 uses ds_read_b64_tr_b16 (hardware transpose read); a scalar
// fallback (TR = false) exists for bring-up and is selected at run time with
// emoasr_set_option("tr_read", 0).
#pragma once
#include <type_traits>
#include "common.h"

__device__ __forceinline__ int c_row(int reg, int lane) {
  return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}
__device__ __forceinline__ int c_col(int lane) { return lane & 31; }

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

// 4 consecutive-k elements (k0..k0+3) for row `row0 + (lane&31)` out of a
// k-major bf16 LDS tile.  Every lane of the wave must execute this (EXEC all 1s).
__device__ __forceinline__ s16x4 tr_read4(const bf16* base, int ld, int k0, int row0, int lane) {
  const int q = (lane & 15) >> 2, p = lane & 3;
  const bf16* addr = base + (k0 + q) * ld + row0 + 16 * ((lane >> 4) & 1) + 4 * p;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(addr));
}

template <typename T> struct Mma;

template <> struct Mma<bf16> {
  static constexpr int KSTEP = 16;
  typedef bf16x8 Frag;

  static __device__ __forceinline__ Frag load_kc(const bf16* base, int ld, int row0, int k0, int lane) {
    return *reinterpret_cast<const bf16x8*>(base + (row0 + (lane & 31)) * ld + k0 + 8 * (lane >> 5));
  }
  template <bool TR>
  static __device__ __forceinline__ Frag load_km(const bf16* base, int ld, int k0, int row0, int lane) {
    const int kb = k0 + 8 * (lane >> 5);
    if constexpr (TR) {
      union { bf16x8 f; s16x4 h[2]; } u;
      u.h[0] = tr_read4(base, ld, kb, row0, lane);
      u.h[1] = tr_read4(base, ld, kb + 4, row0, lane);
      return u.f;
    } else {
      Frag f;
      const bf16* p = base + kb * ld + row0 + (lane & 31);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = p[j * ld];
      return f;
    }
  }
  static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

template <> struct Mma<float> {
  static constexpr int KSTEP = 2;
  typedef float Frag;

  static __device__ __forceinline__ Frag load_kc(const float* base, int ld, int row0, int k0, int lane) {
    return base[(row0 + (lane & 31)) * ld + k0 + (lane >> 5)];
  }
  template <bool TR>
  static __device__ __forceinline__ Frag load_km(const float* base, int ld, int k0, int row0, int lane) {
    return base[(k0 + (lane >> 5)) * ld + row0 + (lane & 31)];
  }
  static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
};

// ---- split products: f32 values as (hi, lo) bf16 pairs, three bf16 MFMAs per product (see csrc/gemm.hip SplitCfg) -------------
//   x ~= hi + lo,  hi = bf16(x),  lo = bf16(x - hi);   a . b ~= a.lo . b.hi + a.hi . b.lo + a.hi . b.hi
// Mma<f32s>: operand tiles in LDS hold one PACKED pair per element (hi in the low, lo in the high 16 bits of the element's 32 bits:
// the f32 tiles' geometry and index arithmetic stay), written once by lds_stage16 when a tile is staged; fragments are 8
// consecutive-k pairs per lane, unpacked with two byte permutes per dword pair.
struct SplitFrag { bf16x8 hi, lo; };
__device__ __forceinline__ unsigned split_pack(float x) {
  const bf16 h = (bf16)x;
  const bf16 l = (bf16)(x - (float)h);
  return (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
}
__device__ __forceinline__ SplitFrag split_unpack8(const unsigned (&w)[8]) {
  union { bf16x8 f; unsigned u[4]; } h, l;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    h.u[m] = __builtin_amdgcn_perm(w[2 * m + 1], w[2 * m], 0x05040100u);   // low halves of the two words
    l.u[m] = __builtin_amdgcn_perm(w[2 * m + 1], w[2 * m], 0x07060302u);   // high halves
  }
  return SplitFrag{h.f, l.f};
}
__device__ __forceinline__ SplitFrag split_regs8(const float (&x)[8]) {
  SplitFrag f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    f.hi[j] = (bf16)x[j];
    f.lo[j] = (bf16)(x[j] - (float)f.hi[j]);
  }
  return f;
}
// staging store of 16 bytes of T into an MFMA operand tile (f32s: packed pairs; every other type: the values themselves)
template <typename T>
__device__ __forceinline__ void lds_stage16(T* dst, const Vec16<T>& v) {
  if constexpr (std::is_same<T, f32s>::value) {
    u32x4 w;
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = split_pack(v.v[e]);
    *reinterpret_cast<u32x4*>(dst) = w;
  } else {
    store16(dst, v);
  }
}

template <> struct Mma<f32s> {
  static constexpr int KSTEP = 16;
  typedef SplitFrag Frag;

  // k-contiguous tile of packed pairs: 8 consecutive k of row row0 + (lane & 31) (two 16-byte reads: ld % 4 == 0, k0 % 8 == 0)
  static __device__ __forceinline__ Frag load_kc(const f32s* base, int ld, int row0, int k0, int lane) {
    const unsigned* p = reinterpret_cast<const unsigned*>(base) + (row0 + (lane & 31)) * ld + k0 + 8 * (lane >> 5);
    const u32x4 a = *reinterpret_cast<const u32x4*>(p), b = *reinterpret_cast<const u32x4*>(p + 4);
    const unsigned w[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return split_unpack8(w);
  }
  template <bool TR>
  static __device__ __forceinline__ Frag load_km(const f32s* base, int ld, int k0, int row0, int lane) {
    const unsigned* p = reinterpret_cast<const unsigned*>(base) + (k0 + 8 * (lane >> 5)) * ld + row0 + (lane & 31);
    unsigned w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = p[j * ld];
    return split_unpack8(w);
  }
  static __device__ __forceinline__ f32x16 mma(const Frag& a, const Frag& b, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, c, 0, 0, 0);
  }
};
