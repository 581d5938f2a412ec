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
