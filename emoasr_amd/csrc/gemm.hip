// MFMA GEMM kernels (gfx950): the contraction workhorse of the encoder.
//
//   gemm_nt : C[M,N]   = epilogue(alpha * A[M,K] . B[N,K]^T)       forward / dgrad
//   gemm_tn : C[N1,N2] (+)= alpha * A[K,N1]^T . B[K,N2]            wgrad (f32 out, split-K)
//
// Replaces the nn.Linear / 1x1 Conv1d / Conv2d calls of the reference:
//   asr/modeling/transformer.py:62-71,94,102-118 (q/k/v/out, FFN w1/w2),
//   asr/modeling/conformer.py:62,103-117 (linear_pos, pointwise convs),
//   asr/modeling/encoders/conv.py:9-19 (Conv2d k3 s2 as implicit GEMM, Linear),
//   asr/modeling/decoders/ctc.py:34,103 (vocabulary head).
//
// Block = 256 threads = 4 waves in a 2x2 arrangement; each wave owns a
// (BM/2)x(BN/2) patch made of 32x32 MFMA tiles.  Tiles are staged global ->
// registers -> LDS (double buffered, one barrier per k-tile); the next k-tile's
// global loads are issued before the current tile's MFMAs.
#include <algorithm>
#include <type_traits>
#include <vector>
#include "mma.h"
#include "../../include/emoasr_hip.h"

namespace {

// ----------------------------------------------------------------------------
// addressing of the "gathered" operand for the Conv2d(k3,s2) implicit GEMM
// rows  m = (b, t2, f2)          -> base offset of the 3x3xC input patch
// cols  k = (kh, kw, c)          -> (kh*F1 + kw)*C + c
// The conv input is channels-last: y1[b, t1, f1, c].
// ----------------------------------------------------------------------------
struct ConvGeom {
  int T1, F1, T2, F2, C;
};
__device__ __forceinline__ long conv_row_base(const ConvGeom& g, int m) {
  const int per_b = g.T2 * g.F2;
  const int b = m / per_b, r = m - b * per_b;
  const int t2 = r / g.F2, f2 = r - t2 * g.F2;
  return (((long)b * g.T1 + 2 * t2) * g.F1 + 2 * f2) * g.C;
}
__device__ __forceinline__ int conv_k_off(const ConvGeom& g, int k) {
  const int p = k / g.C, c = k - p * g.C;
  const int kh = p / 3, kw = p - kh * 3;
  return (kh * g.F1 + kw) * g.C + c;
}

// Data gradient of the same convolution as an implicit GEMM per output-parity class (AMODE == 2):
// dy1[b, t1 = 2i+pt, f1 = 2j+pf, c] = relu'(y1) * sum_{taps (kh,kw) with kh = pt, kw = pf (mod 2)}
//                                       sum_n dy2[b, i - kh/2, j - kw/2, n] * W[n, kh, kw, c]
// rows  m = (b, i, j) of ONE class;  cols k = (tap, n);  A row base = dy2[b, i, j, :], a tap shifts it by
// -(dh*F2 + dw)*C and is zero outside the output map;  B[k][c] = w2r[n][(kh*3+kw)*C + c] (k-major, ld 9C).
// Every dy1 element belongs to exactly one class, so the four launches write dy1 once, with no im2col
// buffer in between (the col2im path wrote and re-read B*T2*F2 x 9C values).
struct DgradGeom {
  int T1, F1, T2, F2, C;
  int pt, pf, nI, nJ, ntap;
  int dh[4], dw[4], wtap[4];  // per tap: source shift (dh, dw) and the tap's column offset inside a w2r row
};
__device__ __forceinline__ void dgrad_row(const DgradGeom& g, int m, int& b, int& i, int& j) {
  const int per_b = g.nI * g.nJ;
  b = m / per_b;
  const int r = m - b * per_b;
  i = r / g.nJ;
  j = r - i * g.nJ;
}

int g_tr_read = 1;
int g_gemm_wholek = 1;   // option "gemm_wholek"
int g_gemm_tile = 0, g_gemm_kb = 0, g_gemm_xcd = 1, g_tn_group_kb = 0, g_tn_place = 0, g_gemm_wide128 = 0;
int g_tn_group_blocks = 0;  // override of a grouped TN launch's block budget (emoasr_set_option "tn_group_blocks"; 0 = auto)
// f32 products as three bf16 MFMAs over (hi, lo) operand pairs (see SplitCfg): asked for PER CALL by the dtype code EMO_F32X3.  The
// entry points of this file note it here for the launch helpers below them (thread-local, for the duration of the call: two engines
// in different modes on two streams, or an autograd thread, cannot see each other's mode)
thread_local int t_f32_split = 0;
struct SplitScope {
  int prev;
  explicit SplitScope(int dtype) : prev(t_f32_split) { t_f32_split = dtype == EMO_F32X3 ? 1 : 0; }
  ~SplitScope() { t_f32_split = prev; }
};
int g_split_tile = 0;       // option "split_tile": tile of the split NT / NN products (1 = 128x128 where it fills the chip twice, 2 = 128x64, 3 = 64x64; 0 = rule)
int g_split_kb = 1;         // option "split_kb": 2 = BK 64 for split reductions of K >= 512
int g_split_min128 = 512;   // option "split_min128": 128 x 128 split tiles from this many tiles on

// XCD-aware block order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  Reading
// the linear id as (xcd, slot) makes XCD x work on ONE contiguous range of the logical block list, so
// blocks that share operand tiles (a band of output rows in NT; one k slice in TN) share an L2.
__device__ __forceinline__ int xcd_remap(int pid, int nblk) {
  const int per = nblk / 8, rem = nblk - per * 8;  // the first `rem` XCDs hold one more block
  const int x = pid % 8, slot = pid / 8;
  return x * per + min(x, rem) + slot;
}

struct NtArgs {
  int M, N, K;
  const void* A; long lda;
  const void* B; long ldb;
  void* C; long ldc;
  emoasr_epilogue_t ep;
  ConvGeom cg;
  DgradGeom dg;
  int nh;  // batched: blockIdx.z = b * nh + h, two-level strides below (elements); 0 = not batched
  long sa_b, sa_h, sb_b, sb_h, sc_b, sc_h;
  int xcd;  // remap block ids so that each XCD works on a contiguous band of rows
};

// KB scales the k extent of a tile: bf16 uses BK = 32 for short reductions (K = 256: fewer, cheaper
// pipeline fill steps) and BK = 64 for long ones (half the barriers per MFMA) -- measured per shape.
template <typename T, int KB = 1> struct TileCfg {
  static constexpr int VEC = 16 / sizeof(T);
  static constexpr int BK = (sizeof(T) == 2 ? 32 : 16) * KB;
  static constexpr int KV = BK / VEC;                       // 16-byte vectors per tile row (= 4)
  static constexpr int LD = BK + (sizeof(T) == 2 ? 8 : 1);  // padded LDS row (elements)
};


// "Split" products (SP; T = float): f32 operands in HBM, every product as THREE bf16 MFMAs over (hi, lo) operand pairs --
//   x = hi + lo,  hi = bf16(x),  lo = bf16(x - hi):   a . b ~= ah . bh + ah . bl + al . bh     (f32 accumulate)
// -- 16 significand bits per operand (relative error ~2^-17 per product, the dropped lo . lo term 2^-18) at 16 / 3 times the rate
// of v_mfma_f32_32x32x2_f32.  The f32 tile is split ONCE per workgroup on its way from the staging registers into LDS (two bf16
// planes per operand, the bf16 kernel's tile geometry); everything outside the k loop -- loads, epilogue, output -- is the f32
// kernel's.  Selected per process by the dtype code EMO_F32X3 (include/emoasr_hip.h): the engine's throughput mode that meets the 1e-3 bar.
template <int KB> struct SplitCfg {
  static constexpr int VEC = 4;                 // floats per 16-byte global vector
  static constexpr int BK = 32 * KB;
  static constexpr int KV = BK / VEC;
  static constexpr int LD = BK + 8;             // bf16 elements per LDS row (k-contiguous planes)
};
__device__ __forceinline__ void split_store4(bf16* hi, bf16* lo, const Vec16<float>& v) {
  bf16x4 h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    h[e] = (bf16)v.v[e];
    l[e] = (bf16)(v.v[e] - (float)h[e]);
  }
  *reinterpret_cast<bf16x4*>(hi) = h;
  *reinterpret_cast<bf16x4*>(lo) = l;
}

#ifndef EMO_GEMM_SWZ
#define EMO_GEMM_SWZ 1
#endif
template <typename T>
__device__ __forceinline__ void lds_store_row(T* dst, const Vec16<T>& v) {
  if constexpr (sizeof(T) == 2) {
    store16(dst, v);  // row stride 80 B keeps 16-B alignment
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[j] = v.v[j];  // odd f32 row stride: scalar stores
  }
}

// BKM = true: B is stored [K][N] (k-major, row stride ldb) -> C = A . B   ("NN", used for dgrad)
//
// Loads are branch-free: out-of-range rows / k-vectors read a clamped in-bounds address and are
// zeroed with a select, so the k-loop has no exec-mask branches.  The epilogue transposes each
// 32x32 accumulator tile through wave-private LDS so that every lane owns 8 consecutive columns
// of one row: bias / saved-activation / residual traffic and the output are 16-byte accesses.
template <typename T, int BM, int BN, int AMODE, bool BKM, bool TR, int KB, bool SP = false>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const NtArgs g) {
  static_assert(!SP || (sizeof(T) == 4 && KB < 8), "split products: f32 operands only");
  using Cfg = TileCfg<T, KB>;
  using M_ = Mma<T>;
  using L = std::conditional_t<SP, bf16, T>;              // element type of the LDS tiles
  constexpr int VEC = SP ? SplitCfg<KB>::VEC : Cfg::VEC, BK = SP ? SplitCfg<KB>::BK : Cfg::BK;
  // SWZ (round 6; bf16, BK = 32): k-contiguous tiles with UNPADDED 64-byte rows whose four 16-byte pieces are XOR-swizzled by
  // (row >> 2) & 3.  With the padded 80-byte rows the two rows of an 8-lane 16-byte store group overlapped in four banks (counters:
  // a third of the LDS cycles of the K = 256 products were bank conflicts, the LDS busy 45 % of the kernel); the swizzled image is
  // conflict-free for the stores (two rows = 128 contiguous bytes) AND for the fragment reads (the four rows of one residue class
  // mod 4 in a 16-lane read group get four different pieces).
  constexpr bool SWZ = EMO_GEMM_SWZ && !SP && sizeof(T) == 2 && KB == 1;
  constexpr int KV = SP ? SplitCfg<KB>::KV : Cfg::KV, LD = SP ? SplitCfg<KB>::LD : (SWZ ? Cfg::BK : Cfg::LD);
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  constexpr int LDBK = BN + (sizeof(L) == 2 ? 32 : 0);  // k-major B tile row stride
  constexpr int BVK = BN / VEC;                          // vectors per k row (k-major B)
  constexpr int A_IT = BM * KV / 256, B_IT = BKM ? BK * BVK / 256 : BN * KV / 256;
  static_assert(A_IT >= 1 && B_IT >= 1, "tile too small for 256 threads");
  constexpr int AS_ELEMS = BM * LD, BS_ELEMS = BKM ? BK * LDBK : BN * LD;
  constexpr int NPL = SP ? 2 : 1;                         // LDS planes per operand (split: hi, lo)
  constexpr int EP_LD = BN + 4;                           // f32 row stride of the staged block tile
  constexpr int NBUF = KB >= 8 ? 1 : 2;   // KB = 8 (whole rows of K = 256 per tile, latency-bound launches): one LDS buffer
  constexpr int STAGE_BYTES = NBUF * NPL * (AS_ELEMS + BS_ELEMS) * (int)sizeof(L);
  constexpr int EPI_ROWS = (BM * BN > 128 * 64) ? BM / 2 : BM;  // the 128x128 tile is staged in two 64-row halves
  constexpr int EPI_BYTES = EPI_ROWS * EP_LD * 4;
  constexpr int SMEM_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;

  __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
  // layout: A planes of every buffer ([buf][plane][AS_ELEMS]), then the B planes likewise
  L* As0 = reinterpret_cast<L*>(smem);
  L* Bs0 = As0 + NBUF * NPL * AS_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;
  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so
  // the linear block id is re-read as (xcd, slot): XCD x then owns a contiguous band of output rows
  // (its A rows are fetched into one L2 only) and walks it n-fastest.
  int bx = blockIdx.x, by = blockIdx.y;
  if (g.xcd) {
    const int lin = xcd_remap(by * gridDim.x + bx, gridDim.x * gridDim.y);
    by = lin / gridDim.x;
    bx = lin - by * gridDim.x;
  }
  const int m0 = by * BM, n0 = bx * BN;
  const T* __restrict__ A = static_cast<const T*>(g.A);
  const T* __restrict__ B = static_cast<const T*>(g.B);
  long c_base = 0;
  if (g.nh > 0) {
    const int bb = blockIdx.z / g.nh, hh = blockIdx.z % g.nh;
    A += bb * g.sa_b + hh * g.sa_h;
    B += bb * g.sb_b + hh * g.sb_h;
    c_base = bb * g.sc_b + hh * g.sc_h;
  }

  // per-thread staging assignments (fixed rows across the k loop): byte offsets into the A / B
  // buffer descriptors (operands are < 4 GiB, checked on the host)
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A), rsB = make_rsrc(B);
  constexpr unsigned SZ = sizeof(T);
  unsigned a_off[A_IT]; bool a_ok[A_IT]; int a_lds[A_IT]; int a_kv[A_IT];
  int a_i[A_IT], a_j[A_IT];  // AMODE == 2: position of the row inside its parity class
  unsigned b_off[B_IT]; bool b_ok[B_IT]; int b_lds[B_IT]; int b_kv[B_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int v = tid + i * 256, r = v / KV, kv = v % KV;
    a_kv[i] = kv * VEC; a_lds[i] = r * LD + (SWZ ? (kv ^ ((r >> 2) & 3)) : kv) * VEC;
    a_ok[i] = (m0 + r) < g.M;
    const int row = a_ok[i] ? (m0 + r) : 0;
    if constexpr (AMODE == 1) a_off[i] = (unsigned)(conv_row_base(g.cg, row) * SZ);
    else if constexpr (AMODE == 2) {
      int bb, ii, jj;
      dgrad_row(g.dg, row, bb, ii, jj);
      a_i[i] = ii; a_j[i] = jj;
      a_off[i] = (unsigned)((((long)bb * g.dg.T2 + ii) * g.dg.F2 + jj) * g.dg.C * SZ);  // may point one row past: guarded per tap
    } else a_off[i] = (unsigned)((long)row * g.lda * SZ);
  }
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int v = tid + i * 256;
    if constexpr (BKM) {
      const int kr = v / BVK, nv = (v % BVK) * VEC;
      b_kv[i] = kr; b_lds[i] = kr * LDBK + nv;
      b_ok[i] = (n0 + nv) < g.N;
      b_off[i] = (unsigned)((n0 + nv) * SZ);
    } else {
      const int r = v / KV, kv = v % KV;
      b_kv[i] = kv * VEC; b_lds[i] = r * LD + (SWZ ? (kv ^ ((r >> 2) & 3)) : kv) * VEC;
      b_ok[i] = (n0 + r) < g.N;
      b_off[i] = (unsigned)((long)(b_ok[i] ? (n0 + r) : 0) * g.ldb * SZ);
    }
  }

  // Register ring, 3 tiles deep: the global loads of tile kt+2 are issued while tile kt is being
  // multiplied and tile kt+1 is copied register -> LDS, so two full k-steps of MFMA + LDS work
  // cover the memory latency (the two-barrier loop waited on vmcnt(0) two thirds of the time).
  struct Stage { Vec16<T> a[A_IT], b[B_IT]; };
  auto load_tile = [&](Stage& r, int k0) {
    // AMODE == 2: a k tile lies inside one tap (C % BK == 0), so the tap's shifts are block-uniform scalars
    int tap = 0, tdh = 0, tdw = 0, tshift = 0, twoff = 0;
    if constexpr (AMODE == 2) {
      tap = min(k0 / g.dg.C, g.dg.ntap - 1);
      tdh = g.dg.dh[tap]; tdw = g.dg.dw[tap];
      tshift = (tdh * g.dg.F2 + tdw) * g.dg.C + tap * g.dg.C;  // subtracted from (row base + k)
      twoff = g.dg.wtap[tap];
    }
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int k = k0 + a_kv[i];
      const bool ok = a_ok[i] && k < g.K;
      unsigned off;
      bool okk = ok;
      if constexpr (AMODE == 1) off = a_off[i] + (unsigned)conv_k_off(g.cg, ok ? k : 0) * SZ;
      else if constexpr (AMODE == 2) {
        const int si = a_i[i] - tdh, sj = a_j[i] - tdw;
        okk = ok && si >= 0 && si < g.dg.T2 && sj >= 0 && sj < g.dg.F2;
        off = a_off[i] + (unsigned)((k - tshift) * (int)SZ);  // k - tap*C = channel n of the source row
      } else off = a_off[i] + (unsigned)k * SZ;
      r.a[i] = buf_load16<T>(rsA, okk ? off : EMO_OOB);
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int k = k0 + b_kv[i];
      const bool ok = b_ok[i] && k < g.K;
      unsigned off;
      if constexpr (AMODE == 2) {
        off = b_off[i] + (unsigned)(((long)(k - tap * g.dg.C) * g.ldb + twoff) * SZ);
      } else off = BKM ? b_off[i] + (unsigned)((long)k * g.ldb * SZ) : b_off[i] + (unsigned)k * SZ;
      r.b[i] = buf_load16<T>(rsB, ok ? off : EMO_OOB);
    }
  };
  auto store_tile = [&](const Stage& r, int buf) {
    L* As = As0 + buf * NPL * AS_ELEMS;
    L* Bs = Bs0 + buf * NPL * BS_ELEMS;
    if constexpr (SP) {
#pragma unroll
      for (int i = 0; i < A_IT; ++i) split_store4(&As[a_lds[i]], &As[AS_ELEMS + a_lds[i]], r.a[i]);
#pragma unroll
      for (int i = 0; i < B_IT; ++i) split_store4(&Bs[b_lds[i]], &Bs[BS_ELEMS + b_lds[i]], r.b[i]);
    } else {
#pragma unroll
      for (int i = 0; i < A_IT; ++i) lds_store_row(&As[a_lds[i]], r.a[i]);
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        if constexpr (BKM) store16(&Bs[b_lds[i]], r.b[i]);
        else lds_store_row(&Bs[b_lds[i]], r.b[i]);
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute_tile = [&](int buf) {
    const L* As = As0 + buf * NPL * AS_ELEMS;
    const L* Bs = Bs0 + buf * NPL * BS_ELEMS;
    if constexpr (SP) {
      using MB = Mma<bf16>;
#pragma unroll
      for (int kk = 0; kk < BK; kk += MB::KSTEP) {
        bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          ah[i] = MB::load_kc(As, LD, wm + i * 32, kk, lane);
          al[i] = MB::load_kc(As + AS_ELEMS, LD, wm + i * 32, kk, lane);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (BKM) {
            bh[j] = MB::template load_km<TR>(Bs, LDBK, kk, wn + j * 32, lane);
            bl[j] = MB::template load_km<TR>(Bs + BS_ELEMS, LDBK, kk, wn + j * 32, lane);
          } else {
            bh[j] = MB::load_kc(Bs, LD, wn + j * 32, kk, lane);
            bl[j] = MB::load_kc(Bs + BS_ELEMS, LD, wn + j * 32, kk, lane);
          }
        }
        // (the two small cross terms first, the leading term last)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = MB::mma(al[i], bh[j], acc[i][j]);
            acc[i][j] = MB::mma(ah[i], bl[j], acc[i][j]);
            acc[i][j] = MB::mma(ah[i], bh[j], acc[i][j]);
          }
      }
    } else {
#pragma unroll
    for (int kk = 0; kk < BK; kk += M_::KSTEP) {
      typename M_::Frag af[TM], bfr[TN];
      // (swizzled tiles: the fragment of k-step kk is piece kk / 8 + (lane >> 5) of row lane & 31 -- block rows are multiples of
      // 32, so the XOR term depends on the lane only)
      auto frag_kc = [&](const L* base, int row0) {
        if constexpr (SWZ) {
          const int piece = (kk / 8 + (lane >> 5)) ^ (((lane & 31) >> 2) & 3);
          return *reinterpret_cast<const typename M_::Frag*>(base + (row0 + (lane & 31)) * LD + piece * 8);
        } else {
          return M_::load_kc(base, LD, row0, kk, lane);
        }
      };
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = frag_kc(As, wm + i * 32);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (BKM) bfr[j] = M_::template load_km<TR>(Bs, LDBK, kk, wn + j * 32, lane);
        else bfr[j] = frag_kc(Bs, wn + j * 32);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = M_::mma(af[i], bfr[j], acc[i][j]);
    }
    }
  };

  const int nk = (g.K + BK - 1) / BK;
  if constexpr (KB >= 8) {
    // Launches of a few workgroups (decoding a few utterances): the time of a 64 x 64 x 256 product is its dependent memory
    // round trips -- the 3-deep ring over eight 32-wide k-tiles waits three to four times.  Here a tile is 256 k wide: all of a
    // K = 256 product's operands are requested at once (eight 16-byte loads per thread and operand) and waited for once; longer
    // reductions prefetch the next 256 into registers while the current one is multiplied.
    Stage sa, sb;
    load_tile(sa, 0);
    for (int kt = 0; kt < nk; kt += 2) {
      if (kt + 1 < nk) load_tile(sb, (kt + 1) * BK);
      if (kt > 0) __syncthreads();
      store_tile(sa, 0);
      __syncthreads();
      compute_tile(0);
      if (kt + 1 < nk) {
        if (kt + 2 < nk) load_tile(sa, (kt + 2) * BK);
        __syncthreads();
        store_tile(sb, 0);
        __syncthreads();
        compute_tile(0);
      }
    }
    __syncthreads();
  } else if constexpr (SP && BM * BN > 128 * 64) {
    // split products on the 128 x 128 tile: ONE register stage (the tile after the one being multiplied), so that the kernel
    // stays under 256 registers and two workgroups share a CU -- with the 3-deep ring it needs 265 and runs one wave per SIMD
    Stage s1;
    load_tile(s1, 0);
    store_tile(s1, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      load_tile(s1, (kt + 1) * BK);
      compute_tile(kt & 1);
      store_tile(s1, (kt + 1) & 1);
      __syncthreads();
    }
  } else {
  Stage s0, s1, s2;  // named stages: static register indexing (a runtime-indexed ring would spill)
  load_tile(s0, 0);
  load_tile(s1, BK);
  store_tile(s0, 0);
  __syncthreads();
  // step(kt): tile kt is in LDS[kt&1]; `nxt` holds tile kt+1 (in flight or landed); `fre` is free
  // loads / stores are issued unconditionally (tiles past K use out-of-bounds buffer offsets and stage zeros):
  // with a static number of loads per step the compiler can wait with a counted vmcnt(N) for the
  // older stage only, instead of draining the stage it has just issued.
  auto step = [&](int kt, Stage& nxt, Stage& fre) {
    load_tile(fre, (kt + 2) * BK);
    compute_tile(kt & 1);
    store_tile(nxt, (kt + 1) & 1);
    __syncthreads();
  };
  for (int kt = 0; kt < nk; kt += 3) {
    step(kt, s1, s2);
    if (kt + 1 < nk) step(kt + 1, s2, s0);
    if (kt + 2 < nk) step(kt + 2, s0, s1);
  }
  }

  // ---- epilogue -------------------------------------------------------------
  // (the loop ended with a barrier: the staging buffers are free to be reused)
  const emoasr_epilogue_t& ep = g.ep;
  const T* __restrict__ res = static_cast<const T*>(ep.residual);
  const T* __restrict__ dpre = static_cast<const T*>(ep.dact_pre);
  T* __restrict__ pre_out = static_cast<T*>(ep.pre_out);
  const bool vec_ok = (g.N % 8 == 0) && (g.ldc % 8 == 0) && (!res || ep.ldr % 8 == 0);
  // The whole BM x BN block tile goes through LDS (f32, row stride BN + 4), then the 256 threads walk it
  // row-major: BN / 8 adjacent lanes cover one full tile row, so every global access of the epilogue
  // (C, residual, dact_pre, pre_out) is a contiguous 2*BN-byte row segment -- full 128-byte lines for
  // BN = 64.  (Per-wave 32-column staging wrote 64-byte half lines and ran the stores of the K = 256
  // products at ~2.3 TB/s: the epilogue cost as much as the k loop.)
  float* sc = reinterpret_cast<float*>(smem);
  constexpr int LPR = BN / 8, RPP = 256 / LPR;  // lanes per tile row, rows per pass
  const int er = tid / LPR, ec = (tid % LPR) * 8;
  const int col = n0 + ec;
#pragma unroll
  for (int half = 0; half < BM / EPI_ROWS; ++half) {
  const int hrow0 = half * EPI_ROWS;
  if (half > 0) __syncthreads();  // the previous half has been read
  auto stage_tile = [&](const f32x16& a, const int i, const int j) __attribute__((always_inline)) {
    const int r0 = wm + i * 32 - hrow0;
    if (r0 >= 0 && r0 < EPI_ROWS) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[(r0 + c_row(r, lane)) * EP_LD + wn + j * 32 + c_col(lane)] = a[r];
    }
  };
  // The epilogue's own operands (residual rows, the saved pre-activation of a backward pass) are fetched BEFORE the tile is staged:
  // the output may alias the residual (x += ...), so behind the barrier the compiler has to keep load -> store -> load order across
  // the passes -- one exposed global round trip per pass and block.  Each thread reads exactly the elements it later writes.
  constexpr int NPASS = EPI_ROWS / RPP;
  constexpr int RAWN = sizeof(T) == 2 ? 1 : 2;
  Vec16<T> pf_res[NPASS][RAWN], pf_dpre[NPASS][RAWN];
  if (vec_ok && (res || dpre)) {
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int row = m0 + hrow0 + pass * RPP + er;
      if (row < g.M && col < g.N) {
        if (res) {
          const T* rp = res + (g.nh > 0 ? c_base : 0) + (long)row * ep.ldr + col;
#pragma unroll
          for (int q = 0; q < RAWN; ++q) pf_res[pass][q] = load16(rp + q * Vec16<T>::N);
        }
        if (dpre) {
          long off = c_base + (long)row * g.ldc + col;
          if constexpr (AMODE == 2) {
            int bb, ii, jj;
            dgrad_row(g.dg, row, bb, ii, jj);
            off = (((long)bb * g.dg.T1 + 2 * ii + g.dg.pt) * g.dg.F1 + 2 * jj + g.dg.pf) * g.dg.C + col;
          }
#pragma unroll
          for (int q = 0; q < RAWN; ++q) pf_dpre[pass][q] = load16(dpre + off + q * Vec16<T>::N);
        }
      }
    }
  }
  auto raw8 = [&](const Vec16<T> (&r)[RAWN], float (&d)[8]) __attribute__((always_inline)) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) d[e] = r[0].get(e);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) { d[e] = r[0].get(e); d[4 + e] = r[RAWN - 1].get(e); }
    }
  };
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) stage_tile(acc[i][j], i, j);
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < EPI_ROWS / RPP; ++pass) {
    const int lrow = pass * RPP + er;
    const int row = m0 + hrow0 + lrow;
    if (row < g.M && col < g.N) {
      float v[8];
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(&sc[lrow * EP_LD + ec]);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(&sc[lrow * EP_LD + ec + 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = v0[e]; v[4 + e] = v1[e]; }
      long off = c_base + (long)row * g.ldc + col;
      if constexpr (AMODE == 2) {
        int bb, ii, jj;
        dgrad_row(g.dg, row, bb, ii, jj);
        off = (((long)bb * g.dg.T1 + 2 * ii + g.dg.pt) * g.dg.F1 + 2 * jj + g.dg.pf) * g.dg.C + col;
      }
      if (vec_ok) {
        if (ep.bias) {
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(ep.bias + col);
          const f32x4 b1 = *reinterpret_cast<const f32x4*>(ep.bias + col + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = ep.alpha * v[e] + b0[e]; v[4 + e] = ep.alpha * v[4 + e] + b1[e]; }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= ep.alpha;
        }
        if (ep.act & EMO_ACT_SAVE_DACT) {
          // the saved tensor is the backward's factor act'(pre) * dropout_scale, not the pre-activation
          float dd[8], mm[8];
          act_dact_vec<8, T>(ep.act & 0xFF, v, dd);
          dropout_mult8(ep.seed, (uint64_t)row * (uint64_t)g.N + col, ep.drop_p, mm);
#pragma unroll
          for (int e = 0; e < 8; ++e) { v[e] *= mm[e]; dd[e] *= mm[e]; }
          if (pre_out) store8<T>(pre_out + off, dd);
        } else {
        if (pre_out) store8<T>(pre_out + off, v);
        act_vec<8, T>(ep.act, v);
        if (dpre) {
          float d[8];
          raw8(pf_dpre[pass], d);
          dact_vec<8, T>(ep.dact, d, v);
        }
        // (vec_ok: N % 8 == 0 and col % 8 == 0, so the 8 mask indices start at an even one)
        dropout_apply8(ep.seed, (uint64_t)row * (uint64_t)g.N + col, ep.drop_p, v);
        }
        if (res) {
          float d[8];
          raw8(pf_res[pass], d);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = d[e] + ep.res_scale * v[e];
        }
        if (ep.out_f32) {
          float* o = static_cast<float*>(g.C) + off;
          *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
        } else {
          store8<T>(static_cast<T*>(g.C) + off, v);
        }
      } else {
        for (int e = 0; e < 8 && col + e < g.N; ++e) {
          float x = ep.alpha * v[e] + (ep.bias ? ep.bias[col + e] : 0.f);
          const float msc = ep.drop_p > 0.f ? dropout_scale(ep.seed, (uint64_t)row * (uint64_t)g.N + col + e, ep.drop_p) : 1.f;
          if (pre_out) pre_out[off + e] = from_f32<T>((ep.act & EMO_ACT_SAVE_DACT) ? apply_dact<T>(ep.act & 0xFF, x) * msc : x);
          x = apply_act<T>(ep.act & 0xFF, x);
          if (dpre) x *= apply_dact<T>(ep.dact, to_f32(dpre[off + e]));
          x *= msc;
          if (res) x = to_f32(res[(g.nh > 0 ? c_base : 0) + (long)row * ep.ldr + col + e]) + ep.res_scale * x;
          if (ep.out_f32) static_cast<float*>(g.C)[off + e] = x;
          else static_cast<T*>(g.C)[off + e] = from_f32<T>(x);
        }
      }
    }
  }
  }
  static_assert(TM * TN <= 4, "at most four 32x32 accumulator tiles per wave");
}

// ----------------------------------------------------------------------------
// TN: C[n1, n2] += alpha * sum_k A[k, n1] * B[k, n2]      (f32 output, atomics
// across split-K slices).  BMODE == 1 gathers B rows through the conv geometry:
// B[k = (b,t2,f2)][n2 = (kh,kw,c)].
// ----------------------------------------------------------------------------
struct TnArgs {
  int N1, N2, K;
  const void* A; long lda;
  const void* B; long ldb;
  float* C; long ldc;
  float alpha;
  int k_tiles_per_split;
  float* colsum;       // optional: colsum[n1] += colsum_scale * sum_k A[k, n1]  (bias gradient)
  float colsum_scale;
  ConvGeom cg;
};

// one (n1 tile, n2 tile, k slice) of a TN product; shared by the plain and the grouped kernels
template <typename T, int BN1, int BN2, int BMODE, bool TR, int KB, bool SP = false>
__device__ __forceinline__ void tn_block(const TnArgs& g, const int bx, const int by, const int bz, const int nbx) {
  static_assert(!SP || sizeof(T) == 4, "split products: f32 operands only");
  using Cfg = TileCfg<T, KB>;
  using M_ = Mma<T>;
  using L = std::conditional_t<SP, bf16, T>;
  constexpr int VEC = Cfg::VEC, BK = SP ? SplitCfg<KB>::BK : Cfg::BK;
  constexpr int PAD = sizeof(L) == 2 ? 32 : 0;
  constexpr int LDA = BN1 + PAD, LDB = BN2 + PAD;
  constexpr int W1 = BN1 / 2, W2 = BN2 / 2, TM = W1 / 32, TN = W2 / 32;
  constexpr int AV = BN1 / VEC, BV = BN2 / VEC;  // vectors per k row
  constexpr int A_IT = BK * AV / 256, B_IT = BK * BV / 256;
  static_assert(A_IT >= 1 && B_IT >= 1, "tile too small");
  constexpr int NPL = SP ? 2 : 1;   // split: hi plane, then lo plane

  __shared__ __attribute__((aligned(16))) L As[2][NPL * BK * LDA];
  __shared__ __attribute__((aligned(16))) L Bs[2][NPL * BK * LDB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int w1 = (wave >> 1) * W1, w2 = (wave & 1) * W2;
  const int n1_0 = by * BN1, n2_0 = bx * BN2;
  const T* __restrict__ A = static_cast<const T*>(g.A);
  const T* __restrict__ B = static_cast<const T*>(g.B);

  const int nk_total = (g.K + BK - 1) / BK;
  const int kt_begin = bz * g.k_tiles_per_split;
  const int kt_end = nk_total < kt_begin + g.k_tiles_per_split ? nk_total : kt_begin + g.k_tiles_per_split;
  if (kt_begin >= kt_end) return;

  int b_koff = 0;
  // conv gather: the rows a thread loads advance by a fixed step (256 / BV rows) from one load to the
  // next, across tiles too, so (b, t2, f2) is carried incrementally -- two integer divisions per block
  // instead of two per 16-byte load (they cost more VALU time than the tile's MFMAs).
  int cv_b = 0, cv_t2 = 0, cv_f2 = 0;
  if constexpr (BMODE == 1) {
    b_koff = conv_k_off(g.cg, n2_0);  // tile lies inside one (kh,kw)
    const int k = kt_begin * BK + tid / BV, per_b = g.cg.T2 * g.cg.F2;
    cv_b = k / per_b;
    const int r = k - cv_b * per_b;
    cv_t2 = r / g.cg.F2;
    cv_f2 = r - cv_t2 * g.cg.F2;
  }
  auto conv_next_row = [&]() -> long {  // row base of the current row, then step to this thread's next row
    const long base = (((long)cv_b * g.cg.T1 + 2 * cv_t2) * g.cg.F1 + 2 * cv_f2) * g.cg.C;
    cv_f2 += 256 / BV;
    while (cv_f2 >= g.cg.F2) { cv_f2 -= g.cg.F2; ++cv_t2; }
    while (cv_t2 >= g.cg.T2) { cv_t2 -= g.cg.T2; ++cv_b; }
    return base;
  };

  struct Stage { Vec16<T> a[A_IT], b[B_IT]; };
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A), rsB = make_rsrc(B);
  constexpr unsigned SZ = sizeof(T);
  auto load_tile = [&](Stage& st, int kt) {
    const int k0 = kt < kt_end ? kt * BK : g.K;  // past this block's k slice: every offset out of bounds -> zeros
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int v = tid + i * 256, kr = v / AV, nv = (v % AV) * VEC;
      const int k = k0 + kr, n = n1_0 + nv;
      const bool ok = k < g.K && n < g.N1;
      st.a[i] = buf_load16<T>(rsA, ok ? (unsigned)(((long)k * g.lda + n) * SZ) : EMO_OOB);
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int v = tid + i * 256, kr = v / BV, nv = (v % BV) * VEC;
      const int k = k0 + kr, n = n2_0 + nv;
      const bool ok = k < g.K && n < g.N2;
      unsigned off;
      if constexpr (BMODE == 1) off = (unsigned)((conv_next_row() + b_koff + nv) * SZ);  // load_tile runs for kt_begin, +1, +2, ... in order
      else off = (unsigned)(((long)k * g.ldb + n) * SZ);
      st.b[i] = buf_load16<T>(rsB, ok ? off : EMO_OOB);
    }
  };
  auto store_tile = [&](const Stage& st, int buf) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int v = tid + i * 256, kr = v / AV, nv = (v % AV) * VEC;
      if constexpr (SP) split_store4(&As[buf][kr * LDA + nv], &As[buf][BK * LDA + kr * LDA + nv], st.a[i]);
      else store16(&As[buf][kr * LDA + nv], st.a[i]);
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int v = tid + i * 256, kr = v / BV, nv = (v % BV) * VEC;
      if constexpr (SP) split_store4(&Bs[buf][kr * LDB + nv], &Bs[buf][BK * LDB + kr * LDB + nv], st.b[i]);
      else store16(&Bs[buf][kr * LDB + nv], st.b[i]);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // bias gradient: the nbx blocks of one row of tiles hold the same A tile; they take its k-tiles in turn, so that the blocks that
  // share operand tiles do the same amount of work and stay within an L2 lifetime of each other (with the column sums on the bx = 0
  // blocks alone those fell behind their neighbours and the 1024x256 product re-read a third of its A operand from HBM)
  const bool do_colsum = g.colsum != nullptr && tid < BN1;
  float csum = 0.f;
  // same 3-deep register ring as the NT kernel (see there)
  Stage s0, s1, s2;
  load_tile(s0, kt_begin);
  load_tile(s1, kt_begin + 1);
  store_tile(s0, 0);
  __syncthreads();
  auto step = [&](int kt, Stage& nxt, Stage& fre) {
    const int buf = (kt - kt_begin) & 1;
    load_tile(fre, kt + 2);
    if (do_colsum && kt % nbx == bx) {
#pragma unroll
      for (int k = 0; k < BK; ++k) {
        if constexpr (SP) csum += (float)As[buf][k * LDA + tid] + (float)As[buf][BK * LDA + k * LDA + tid];
        else csum += to_f32(As[buf][k * LDA + tid]);
      }
    }
    if constexpr (SP) {
      using MB = Mma<bf16>;
#pragma unroll
      for (int kk = 0; kk < BK; kk += MB::KSTEP) {
        bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          ah[i] = MB::template load_km<TR>(As[buf], LDA, kk, w1 + i * 32, lane);
          al[i] = MB::template load_km<TR>(As[buf] + BK * LDA, LDA, kk, w1 + i * 32, lane);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          bh[j] = MB::template load_km<TR>(Bs[buf], LDB, kk, w2 + j * 32, lane);
          bl[j] = MB::template load_km<TR>(Bs[buf] + BK * LDB, LDB, kk, w2 + j * 32, lane);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = MB::mma(al[i], bh[j], acc[i][j]);
            acc[i][j] = MB::mma(ah[i], bl[j], acc[i][j]);
            acc[i][j] = MB::mma(ah[i], bh[j], acc[i][j]);
          }
      }
    } else {
#pragma unroll
    for (int kk = 0; kk < BK; kk += M_::KSTEP) {
      typename M_::Frag af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = M_::template load_km<TR>(As[buf], LDA, kk, w1 + i * 32, lane);
#pragma unroll
      for (int j = 0; j < TN; ++j) bfr[j] = M_::template load_km<TR>(Bs[buf], LDB, kk, w2 + j * 32, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = M_::mma(af[i], bfr[j], acc[i][j]);
    }
    }
    store_tile(nxt, buf ^ 1);
    __syncthreads();
  };
  for (int kt = kt_begin; kt < kt_end; kt += 3) {
    step(kt, s1, s2);
    if (kt + 1 < kt_end) step(kt + 1, s2, s0);
    if (kt + 2 < kt_end) step(kt + 2, s0, s1);
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n2_0 + w2 + j * 32 + c_col(lane);
      if (col >= g.N2) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = n1_0 + w1 + i * 32 + c_row(r, lane);
        if (row >= g.N1) continue;
        atomicAdd(&g.C[(long)row * g.ldc + col], g.alpha * acc[i][j][r]);
      }
    }
  if (do_colsum && n1_0 + tid < g.N1) atomicAdd(&g.colsum[n1_0 + tid], g.colsum_scale * csum);
}

template <typename T, int BN1, int BN2, int BMODE, bool TR, int KB, bool SP = false>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const TnArgs g) {
  // all (n1, n2) tiles of one k slice read the same rows of A and B: keep a slice on one XCD
  const int gx = gridDim.x, gy = gridDim.y;
  const int lin = xcd_remap((blockIdx.z * gy + blockIdx.y) * gx + blockIdx.x, gx * gy * gridDim.z);
  tn_block<T, BN1, BN2, BMODE, TR, KB, SP>(g, lin % gx, (lin / gx) % gy, lin / (gx * gy), gx);
}

// Grouped form: up to EMOASR_TN_GROUP_MAX independent products in one launch (the ~10 weight
// gradients of one encoder layer: each alone is 16..64 tiles, far too few for 256 CUs).  Blocks
// are numbered problem by problem; start[p] is the first block of problem p.
// Placement (option "tn_place", off by default): a k slice of one problem (all its (n1, n2) tiles read the same rows of A and B) is
// the unit that has to share an L2.  The hardware hands block i to XCD i % 8; with placement the host deals WHOLE slices to the eight
// XCDs (largest first, to the least loaded) and block i = (xcd, slot) looks its slice up in that XCD's list.  The default --
// contiguous eighths of the problem-by-problem block list -- cuts slices at the XCD boundaries.  Measured at 35 k rows
// (tools/tn_probe.py, tools/tn_ab.sh): alone in a loop the placed launch of a layer's nine products is ahead (182 vs 193 us, and
// with BK = 64 it read 790 instead of 855 MB), but inside the training step it is behind (291 vs 206 us per layer, rocprofv3
// kernel trace of bench.py under both settings), so the step keeps the contiguous mapping.
constexpr int TN_PLACE_MAX = 24;
struct TnPlace {
  unsigned short start[8][TN_PLACE_MAX + 1];   // first slot of entry e on XCD x; start[x][cnt[x]] = blocks of XCD x
  unsigned char prob[8][TN_PLACE_MAX], split[8][TN_PLACE_MAX];
  unsigned char cnt[8];
};
struct TnGroup {
  int n;
  int xcd;   // 1: slices placed per XCD (place), 2: contiguous eighths of the block list, 0: hardware block order
  int start[EMOASR_TN_GROUP_MAX + 1];
  TnArgs p[EMOASR_TN_GROUP_MAX];
  TnPlace place;
};
template <typename T, bool TR, int KB, int BT, bool SP = false>
__global__ __launch_bounds__(256) void gemm_tn_grouped_kernel(const TnGroup G) {
  if (G.xcd == 1) {
    const int x = blockIdx.x & 7, slot = blockIdx.x >> 3, cnt = G.place.cnt[x];
    if (slot >= G.place.start[x][cnt]) return;
    int e = 0;
    while (e + 1 < cnt && slot >= G.place.start[x][e + 1]) ++e;
    const TnArgs& g = G.p[G.place.prob[x][e]];
    const int local = slot - G.place.start[x][e];
    const int tx = (g.N2 + BT - 1) / BT;
    tn_block<T, BT, BT, 0, TR, KB, SP>(g, local % tx, local / tx, G.place.split[x][e], tx);
    return;
  }
  const int bid = G.xcd ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;  // see gemm_tn_kernel
  int p = 0;
  while (p + 1 < G.n && bid >= G.start[p + 1]) ++p;
  const TnArgs& g = G.p[p];
  const int local = bid - G.start[p];
  const int tx = (g.N2 + BT - 1) / BT, ty = (g.N1 + BT - 1) / BT;
  tn_block<T, BT, BT, 0, TR, KB, SP>(g, local % tx, (local / tx) % ty, local / (tx * ty), tx);
}

template <typename T, int AMODE, bool BKM, bool TR, bool SP = false>
int launch_nt_(const NtArgs& a_in, hipStream_t s, int nz = 1) {
  NtArgs a = a_in;
  a.xcd = g_gemm_xcd;
  if constexpr (SP) {
    // split products: the bf16 kernel's tile rule (128x64 while that fills the chip 1.5 times, else 64x64), BK = 32
    const long t12864 = (long)cdiv(a.M, 128) * cdiv(a.N, 64) * nz, t128 = (long)cdiv(a.M, 128) * cdiv(a.N, 128) * nz;
    // 64x64 everywhere (four workgroups per CU): measured inside the f32x3 training step against the bf16 kernel's rule (128x64
    // where that fills the chip 1.5 times: 17.17 ms), 128x128 tiles (17.27) and BK = 64 (18.35): 17.00 ms per step
    // ... and 128 x 128 with ONE register stage (two workgroups per CU; with the 3-deep ring it ran one wave per SIMD and lost)
    // where that tile still fills the chip: 16.70 against 17.12 ms per step
    (void)t12864;
    int tile = (t128 >= g_split_min128 && a.N % 128 == 0) ? 1 : 3;
    if (g_split_tile == 1 && a.N % 128 == 0) tile = 1;
    else if (g_split_tile >= 2) tile = g_split_tile;
    const int kb = (g_split_kb == 2 && a.K >= 512) ? 2 : 1;
#define EMO_SP_LAUNCH(BM_, BN_)                                                            \
  do {                                                                                     \
    dim3 grid(cdiv(a.N, BN_), cdiv(a.M, BM_), nz);                                         \
    if (kb == 2) gemm_nt_kernel<T, BM_, BN_, AMODE, BKM, TR, 2, true><<<grid, 256, 0, s>>>(a); \
    else gemm_nt_kernel<T, BM_, BN_, AMODE, BKM, TR, 1, true><<<grid, 256, 0, s>>>(a);      \
  } while (0)
    if (tile == 1) EMO_SP_LAUNCH(128, 128);
    else if (tile == 2) EMO_SP_LAUNCH(128, 64);
    else EMO_SP_LAUNCH(64, 64);
#undef EMO_SP_LAUNCH
    EMO_LAUNCH_CHECK();
    return 0;
  }
  // Tile: 128x64 once that still gives >= 384 blocks (1.5 per CU), else 64x64.  (A 128x128 tile
  // was measured slower than 128x64 on every shape of the L2 model.)  k extent: BK = 64 for long
  // bf16 reductions, 32 for K = 256.  g_gemm_tile / g_gemm_kb: tuning overrides (emoasr_set_option).
  const long t12864 = (long)cdiv(a.M, 128) * cdiv(a.N, 64) * nz;
  // ... and 64x64 again for the stacked row counts (M >= 16 k): with four to five rounds of blocks either way the smaller tile's
  // higher occupancy wins -- every instantiation of the training step 4-7 % faster in the kernel trace (6.92 -> 6.51 ms per step)
  // (round 4, same shapes one by one: the 128-row tile is the faster one where the output is wide -- N >= 512: ffn1 45 / 50 us, qkv
  // 36 / 43, d_ffn2 38 / 44 -- and the slower one for N = 256 -- ffn2 47 / 44, out 18.5 / 17.2.  INSIDE the training step the mixed
  // rule is slower, three A/B pairs on one box: 29.89 against 29.65 ms per step -- option "gemm_wide128", off)
  // (round 6, after the attention kernels stopped hammering the CUs' vector-memory paths: the mixed rule IS faster inside the step --
  // 128-row tiles from N = 768 on, four A/B pairs on one box: 26.85 against 26.94 ms (N >= 512), 26.96 against 27.10 (all shapes);
  // one by one at 35 k rows: ffn1 with its epilogue 75.9 -> 72.6 us, qkv 40.5 -> 36.6, d_ffn2 48.8 -> 43.3, N = 256 products 8 % slower)
  const bool stacked64 = AMODE == 0 && a.M >= 16384 && !(a.N >= (g_gemm_wide128 ? 512 : 768));
  const int tile = g_gemm_tile ? g_gemm_tile : (stacked64 ? 3 : (t12864 >= 384 ? 2 : 3));
  const int kb = sizeof(T) == 2 ? (g_gemm_kb ? g_gemm_kb : (a.K >= 512 ? 2 : 1)) : 1;
#define EMO_NT_LAUNCH(BM_, BN_)                                                           \
  do {                                                                                    \
    dim3 grid(cdiv(a.N, BN_), cdiv(a.M, BM_), nz);                                        \
    if constexpr (sizeof(T) == 2) {                                                       \
      if (kb == 2) gemm_nt_kernel<T, BM_, BN_, AMODE, BKM, TR, 2><<<grid, 256, 0, s>>>(a); \
      else gemm_nt_kernel<T, BM_, BN_, AMODE, BKM, TR, 1><<<grid, 256, 0, s>>>(a);         \
    } else {                                                                              \
      gemm_nt_kernel<T, BM_, BN_, AMODE, BKM, TR, 1><<<grid, 256, 0, s>>>(a);              \
    }                                                                                     \
  } while (0)
  if constexpr (sizeof(T) == 2 && AMODE == 0 && !BKM) {
    // a few workgroups only (decoding): 256-wide k tiles, one wait per 256 of K (see the kernel)
    if (g_gemm_wholek && a.M <= 2048 && a.K % 256 == 0 && nz == 1) {
      dim3 grid(cdiv(a.N, 64), cdiv(a.M, 64), 1);
      gemm_nt_kernel<T, 64, 64, AMODE, BKM, TR, 8><<<grid, 256, 0, s>>>(a);
      EMO_LAUNCH_CHECK();
      return 0;
    }
  }
  // (64 x 256 tiles -- wave tile 32 x 128, an A row block fetched once per 256 output columns -- were measured at 35 k rows in
  // round 4: 10-30 % SLOWER on every shape, e.g. ffn2 51.7 against 44.4 us, d_ffn1 44.0 / 38.0, out 22.8 / 17.2: two workgroups
  // per CU instead of seven; these products are bound by latency under low occupancy, not by L1 fill traffic.  Removed.)
  if (tile == 1) EMO_NT_LAUNCH(128, 128);
  else if (tile == 2) EMO_NT_LAUNCH(128, 64);
  else EMO_NT_LAUNCH(64, 64);
#undef EMO_NT_LAUNCH
  EMO_LAUNCH_CHECK();
  return 0;
}
// f32 storage with dtype EMO_F32X3: the split kernels (SP); everything else as before
template <typename T> constexpr bool is_f32 = sizeof(T) == 4;
template <typename T, int AMODE>
int launch_nt(const NtArgs& a, hipStream_t s) {
  if constexpr (is_f32<T>) { if (t_f32_split) return launch_nt_<T, AMODE, false, true, true>(a, s); }
  return launch_nt_<T, AMODE, false, true>(a, s);
}
template <typename T>
int launch_nn(const NtArgs& a, hipStream_t s, int nz = 1) {
  if constexpr (is_f32<T>) { if (t_f32_split) return launch_nt_<T, 0, true, true, true>(a, s, nz); }
  return g_tr_read ? launch_nt_<T, 0, true, true>(a, s, nz) : launch_nt_<T, 0, true, false>(a, s, nz);
}

template <typename T, int BMODE, bool SP = false>
int launch_tn(TnArgs a, hipStream_t s) {
  if constexpr (is_f32<T> && !SP) { if (t_f32_split) return launch_tn<T, BMODE, true>(a, s); }
  const bool big = (long)cdiv(a.N1, 128) * cdiv(a.N2, 128) >= 32 && a.N1 >= 128 && a.N2 >= 128;
  // BK = 32 with the 128x128 tile (three resident blocks per CU instead of two; see emoasr_gemm_tn_grouped), 64 for long reductions on 64x64
  const int kb = sizeof(T) == 2 ? (g_gemm_kb ? g_gemm_kb : (a.K >= 512 && !big ? 2 : 1)) : 1;
  const int BK = (SP ? 32 : TileCfg<T>::BK) * kb;
  const int nk = cdiv(a.K, BK);
  const int bn = big ? 128 : 64;
  const long tiles = (long)cdiv(a.N1, bn) * cdiv(a.N2, bn);
  // split-K: as many slices as fit ONE round of resident blocks (rounding up past it leaves a mostly empty
  // second round) with at least 4 k-tiles each, but keep
  // the f32 atomic traffic (output bytes x slices) around 8 MB: global float atomics run at
  // ~1.3 TB/s chip-wide, so more slices than that make the kernel atomic-bound.
  const long slots_ = g_tn_group_blocks > 0 ? g_tn_group_blocks : ((big && kb == 2) || (big && SP) ? 512 : 768);   // (split, 128 tile: 80 KB of LDS)
  int splits = (int)std::max(1L, slots_ / tiles);  // one full round of resident blocks (2 per CU for the 128x128 BK=64 tile, else 3)
  const long out_bytes = (long)a.N1 * a.N2 * 4;
  // ... unless the reduction is so long that the atomics stay below ~10 % of the product's own time
  // (estimated at 300 TFLOP/s): the Conv2d weight gradient (K = B*T'*F2 ~ 130 k) wants 15 slices, not 3
  const double est_s = 2.0 * a.N1 * a.N2 * (double)a.K / 300e12;
  const long cap_work = (long)(0.25 * est_s * 1.3e12 / (double)out_bytes);
  const int cap = (int)std::max(std::max(1L, (8L << 20) / out_bytes), cap_work);
  const int floor_splits = (int)std::min((long)splits, (128 + tiles - 1) / tiles);  // never starve the chip
  splits = std::max(floor_splits, std::min(splits, cap));
  splits = std::max(1, std::min(splits, nk / 4 > 0 ? nk / 4 : 1));
  a.k_tiles_per_split = cdiv(nk, splits);
  splits = cdiv(nk, a.k_tiles_per_split);
  dim3 grid(cdiv(a.N2, bn), cdiv(a.N1, bn), splits);
#define EMO_TN_LAUNCH(BN_, TR_)                                                     \
  do {                                                                              \
    if constexpr (sizeof(T) == 2) {                                                 \
      if (kb == 2) gemm_tn_kernel<T, BN_, BN_, BMODE, TR_, 2><<<grid, 256, 0, s>>>(a); \
      else gemm_tn_kernel<T, BN_, BN_, BMODE, TR_, 1><<<grid, 256, 0, s>>>(a);         \
    } else {                                                                        \
      gemm_tn_kernel<T, BN_, BN_, BMODE, TR_, 1, SP><<<grid, 256, 0, s>>>(a);          \
    }                                                                               \
  } while (0)
  if constexpr (SP) {
    if (big) EMO_TN_LAUNCH(128, true);
    else EMO_TN_LAUNCH(64, true);
  } else {
  if (big) {
    if (g_tr_read) EMO_TN_LAUNCH(128, true);
    else EMO_TN_LAUNCH(128, false);
  } else {
    if (g_tr_read) EMO_TN_LAUNCH(64, true);
    else EMO_TN_LAUNCH(64, false);
  }
  }
#undef EMO_TN_LAUNCH
  EMO_LAUNCH_CHECK();
  return 0;
}

}  // namespace

int emo_conv_big_enabled();
bool emo_gemm_nt_big_wants(int M, int N, int K, long lda, long ldb, long ldc, const emoasr_epilogue_t& ep);
int emo_gemm_nt_big_ep(int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                       const emoasr_epilogue_t& ep, hipStream_t s);
int emo_conv2_fwd_big(int B, int T1, int F1, int C, const void* y1, const void* w, void* y2, const float* bias,
                      int relu, hipStream_t s);

void emo_gemm_set_tr_read(int v) { g_tr_read = v; }
void emo_gemm_set_tile(int v) { g_gemm_tile = v; }
void emo_gemm_set_wide128(int v) { g_gemm_wide128 = v ? 1 : 0; }
void emo_gemm_set_tn_group_blocks(int v) { g_tn_group_blocks = v > 0 ? v : 0; }
void emo_gemm_set_kb(int v) { g_gemm_kb = v; }
void emo_gemm_set_wholek(int v) { g_gemm_wholek = v; }
void emo_gemm_set_tn_place(int v) { g_tn_place = v != 0; }
void emo_gemm_set_tn_group_kb(int v) { g_tn_group_kb = (v == 1 || v == 2) ? v : 0; }
void emo_gemm_set_xcd(int v) { g_gemm_xcd = v; }
void emo_gemm_set_split_tile(int v) { g_split_tile = (v >= 1 && v <= 3) ? v : 0; }
void emo_gemm_set_split_kb(int v) { g_split_kb = v == 2 ? 2 : 1; }
void emo_gemm_set_split_min128(int v) { g_split_min128 = v > 0 ? v : 512; }

static int check_vec(long ld, int dtype, const char* what) {
  const int vec = dtype == EMO_BF16 ? 8 : 4;
  if (ld % vec != 0) { emo_set_error("gemm: %s=%ld must be a multiple of %d elements", what, ld, vec); return 1; }
  return 0;
}

extern "C" int emoasr_gemm_nt(int dtype, int M, int N, int K, const void* A, long lda, const void* B,
                              long ldb, void* C, long ldc, const emoasr_epilogue_t* ep, void* stream) {
  SplitScope split_scope(dtype);
  EMO_CHECK(M > 0 && N > 0 && K > 0, "gemm_nt: empty problem %d %d %d", M, N, K);
  if (check_vec(lda, dtype, "lda") || check_vec(ldb, dtype, "ldb") || check_vec(K, dtype, "K")) return 1;
  NtArgs a{};
  a.M = M; a.N = N; a.K = K; a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc;
  if (ep) a.ep = *ep; else { a.ep = emoasr_epilogue_t{}; a.ep.alpha = 1.f; }
  const double esz_ = dtype == EMO_BF16 ? 2.0 : 4.0;   // algorithmic work of the launch, for the family timer (bench.py)
  EmoTimerScope timer_(EMO_TIMER_GEMM_NT_NN, (hipStream_t)stream, 2.0 * M * N * K,
                       ((double)M * K + (double)N * K + (double)M * N * (1 + (a.ep.residual ? 1 : 0) + (a.ep.pre_out ? 1 : 0))) * esz_);
  // wide bf16 products over many rows (q/k/v, feed-forward w1, pointwise conv 1): the large-tile kernel of gemm_big.hip
  if (dtype == EMO_BF16 && emo_gemm_nt_big_wants(M, N, K, lda, ldb, ldc, a.ep))
    return emo_gemm_nt_big_ep(M, N, K, A, lda, B, ldb, C, ldc, a.ep, (hipStream_t)stream);
  EMO_DISPATCH(dtype, return (launch_nt<T, 0>(a, (hipStream_t)stream)));
}

// C[M,N] = epilogue(A[M,K] . B[K,N])  -- B k-major; dgrad: dX = dY . W with W stored [out,in].
extern "C" int emoasr_gemm_nn(int dtype, int M, int N, int K, const void* A, long lda, const void* B,
                              long ldb, void* C, long ldc, const emoasr_epilogue_t* ep, void* stream) {
  SplitScope split_scope(dtype);
  EMO_CHECK(M > 0 && N > 0 && K > 0, "gemm_nn: empty problem %d %d %d", M, N, K);
  if (check_vec(lda, dtype, "lda") || check_vec(ldb, dtype, "ldb") || check_vec(K, dtype, "K") ||
      check_vec(N, dtype, "N")) return 1;
  NtArgs a{};
  a.M = M; a.N = N; a.K = K; a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc;
  if (ep) a.ep = *ep; else { a.ep = emoasr_epilogue_t{}; a.ep.alpha = 1.f; }
  const double esz_ = dtype == EMO_BF16 ? 2.0 : 4.0;
  EmoTimerScope timer_(EMO_TIMER_GEMM_NT_NN, (hipStream_t)stream, 2.0 * M * N * K,
                       ((double)M * K + (double)N * K + (double)M * N * (1 + (a.ep.residual ? 1 : 0) + (a.ep.dact_pre ? 1 : 0))) * esz_);
  EMO_DISPATCH(dtype, return (launch_nn<T>(a, (hipStream_t)stream)));
  return 0;
}

// Batched C[b,h] = alpha * A[b,h] . B[b,h] (B k-major), two-level batch strides in elements.
// Used by the attention backward: dV = Pd^T-stored . dO and dK = dS^T-stored . Q per (batch, head).
extern "C" int emoasr_gemm_nn_batched(int dtype, int M, int N, int K, const void* A, long lda, long sa_b,
                                      long sa_h, const void* B, long ldb, long sb_b, long sb_h, void* C,
                                      long ldc, long sc_b, long sc_h, int nb, int nh, float alpha,
                                      int accumulate, void* stream) {
  SplitScope split_scope(dtype);
  EMO_CHECK(M > 0 && N > 0 && K > 0 && nb > 0 && nh > 0, "gemm_nn_batched: empty problem");
  // K need not be a multiple of the vector width: rows of A are padded up to lda (zero-filled)
  if (check_vec(lda, dtype, "lda") || check_vec(ldb, dtype, "ldb") ||
      check_vec(N, dtype, "N") || check_vec(sa_b, dtype, "sa_b") || check_vec(sa_h, dtype, "sa_h") ||
      check_vec(sb_b, dtype, "sb_b") || check_vec(sb_h, dtype, "sb_h")) return 1;
  NtArgs a{};
  a.M = M; a.N = N; a.K = K; a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc;
  a.ep = emoasr_epilogue_t{}; a.ep.alpha = alpha;
  if (accumulate) { a.ep.residual = C; a.ep.ldr = (int)ldc; a.ep.res_scale = 1.f; }  // C += (batch offset applied in-kernel)
  a.nh = nh; a.sa_b = sa_b; a.sa_h = sa_h; a.sb_b = sb_b; a.sb_h = sb_h; a.sc_b = sc_b; a.sc_h = sc_h;
  EMO_DISPATCH(dtype, return (launch_nn<T>(a, (hipStream_t)stream, nb * nh)));
  return 0;
}

extern "C" int emoasr_gemm_tn(int dtype, int N1, int N2, int K, const void* A, long lda, const void* B,
                              long ldb, float* C, long ldc, float alpha, int accumulate, float* colsum,
                              float colsum_scale, void* stream) {
  SplitScope split_scope(dtype);
  EMO_CHECK(N1 > 0 && N2 > 0 && K > 0, "gemm_tn: empty problem");
  // N1 may be ragged when lda is padded (rows readable up to lda)
  if (check_vec(lda, dtype, "lda") || check_vec(ldb, dtype, "ldb") || check_vec(N2, dtype, "N2")) return 1;
  EMO_CHECK(N1 % (dtype == EMO_BF16 ? 8 : 4) == 0 || lda >= (N1 + 7) / 8 * 8, "gemm_tn: ragged N1 needs padded lda");
  if (!accumulate) {
    if (ldc == N2) hipMemsetAsync(C, 0, sizeof(float) * (size_t)N1 * N2, (hipStream_t)stream);
    else hipMemset2DAsync(C, sizeof(float) * ldc, 0, sizeof(float) * N2, N1, (hipStream_t)stream);
    if (colsum) hipMemsetAsync(colsum, 0, sizeof(float) * N1, (hipStream_t)stream);
  }
  TnArgs a{};
  a.N1 = N1; a.N2 = N2; a.K = K; a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc;
  a.alpha = alpha; a.colsum = colsum; a.colsum_scale = colsum_scale;
  EmoTimerScope timer_(EMO_TIMER_GEMM_TN, (hipStream_t)stream, 2.0 * N1 * N2 * K,
                       ((double)K * N1 + (double)K * N2) * (dtype == EMO_BF16 ? 2.0 : 4.0) + 4.0 * N1 * N2);
  EMO_DISPATCH(dtype, return (launch_tn<T, 0>(a, (hipStream_t)stream)));
}

// Grouped weight-gradient products (always accumulating): see TnGroup.
extern "C" int emoasr_gemm_tn_grouped(int dtype, int n, const emoasr_tn_problem_t* probs, void* stream) {
  EMO_CHECK(n > 0 && n <= EMOASR_TN_GROUP_MAX, "gemm_tn_grouped: n=%d outside 1..%d", n, EMOASR_TN_GROUP_MAX);
  TnGroup G{};
  G.n = n;
  long tiles = 0;
  // 128x128 tiles (16 MFMAs per wave and k step instead of 4) when every product is at least that large
  int bt = 128;
  for (int i = 0; i < n; ++i)
    if (probs[i].N1 < 128 || probs[i].N2 < 128) bt = 64;
  if (g_gemm_tile == 3) bt = 64;
  // k extent of a tile: 32 for the 128x128 bf16 tile (40 KB of LDS, three blocks per CU), 64 for the 64x64 one.  (BK = 64 with the
  // 128 tile is 80 KB: the 480-block launch of a layer then ran in TWO rounds, blocks sharing operand tiles were no longer
  // co-resident and the launch read 1.2-1.6x its operands from HBM; BK = 32 reads them once -- 216 -> 185 us at 35 k rows,
  // tools/tn_probe.py.)  Option "gemm_kb" overrides.
  const int kb = dtype == EMO_BF16 ? (g_tn_group_kb ? g_tn_group_kb : g_gemm_kb ? g_gemm_kb : (bt == 128 ? 1 : 2)) : 1;
  const bool split = dtype == EMO_F32X3;
  const int BK = (dtype == EMO_BF16 || split ? 32 : 16) * kb;
  for (int i = 0; i < n; ++i) {
    const emoasr_tn_problem_t& q = probs[i];
    EMO_CHECK(q.N1 > 0 && q.N2 > 0 && q.K > 0, "gemm_tn_grouped: empty problem %d", i);
    if (check_vec(q.lda, dtype, "lda") || check_vec(q.ldb, dtype, "ldb") || check_vec(q.N2, dtype, "N2")) return 1;
    EMO_CHECK(q.N1 % (dtype == EMO_BF16 ? 8 : 4) == 0 || q.lda >= (q.N1 + 7) / 8 * 8,
              "gemm_tn_grouped: ragged N1 needs padded lda");
    tiles += (long)cdiv(q.N1, bt) * cdiv(q.N2, bt);
  }
  // at least 4 k-tiles per slice, and per problem no more f32 atomic traffic than ~8 MB (see launch_tn)
  // one split factor for the whole group, chosen so that the launch is ONE full round of resident blocks:
  // three blocks per CU for the 128x128 BK=32 tile (40 KB of LDS) and the 64x64 one (48 KB), two for 128x128 BK=64.  Rounding
  // the block count up past that measured slower every time (at 35 k rows: 768 blocks 185 us, 896 blocks 257 us).
  const long slots = g_tn_group_blocks > 0 ? g_tn_group_blocks : ((bt == 128 && kb == 2) || (bt == 128 && split) ? 512 : 768);
  const int want = (int)std::max(1L, slots / tiles);
  int start = 0;
  for (int i = 0; i < n; ++i) {
    const emoasr_tn_problem_t& q = probs[i];
    TnArgs& a = G.p[i];
    a.N1 = q.N1; a.N2 = q.N2; a.K = q.K; a.A = q.A; a.lda = q.lda; a.B = q.B; a.ldb = q.ldb;
    a.C = q.C; a.ldc = q.ldc; a.alpha = q.alpha; a.colsum = q.colsum; a.colsum_scale = q.colsum_scale;
    const int nk = cdiv(q.K, BK);
    const int cap = (int)std::max(1L, (8L << 20) / ((long)q.N1 * q.N2 * 4));
    int splits = std::max(1, std::min(std::min(want, cap), nk / 4 > 0 ? nk / 4 : 1));
    a.k_tiles_per_split = cdiv(nk, splits);
    splits = cdiv(nk, a.k_tiles_per_split);
    G.start[i] = start;
    start += cdiv(q.N1, bt) * cdiv(q.N2, bt) * splits;
  }
  G.start[n] = start;
  hipStream_t s = (hipStream_t)stream;
  G.xcd = g_tn_place ? 1 : (g_gemm_xcd ? 2 : 0);
  if (G.xcd == 1) {   // deal whole slices to the XCDs: largest first, each to the XCD with the fewest blocks so far
    struct Slice { int tiles, prob, split; };
    std::vector<Slice> sl;
    for (int i = 0; i < n; ++i) {
      const int tiles_i = cdiv(probs[i].N1, bt) * cdiv(probs[i].N2, bt);
      const int splits_i = (G.start[i + 1] - G.start[i]) / tiles_i;
      for (int z = 0; z < splits_i; ++z) sl.push_back({tiles_i, i, z});
    }
    std::stable_sort(sl.begin(), sl.end(), [](const Slice& a, const Slice& b) { return a.tiles > b.tiles; });
    int load[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool fits = true;
    for (const Slice& q : sl) {
      int x = 0;
      for (int j = 1; j < 8; ++j)
        if (load[j] < load[x]) x = j;
      if (cnt[x] == TN_PLACE_MAX || load[x] + q.tiles > 65535 || q.split > 255) { fits = false; break; }
      G.place.start[x][cnt[x]] = (unsigned short)load[x];
      G.place.prob[x][cnt[x]] = (unsigned char)q.prob;
      G.place.split[x][cnt[x]] = (unsigned char)q.split;
      load[x] += q.tiles;
      ++cnt[x];
    }
    if (fits) {
      int most = 0;
      for (int x = 0; x < 8; ++x) {
        G.place.start[x][cnt[x]] = (unsigned short)load[x];
        G.place.cnt[x] = (unsigned char)cnt[x];
        most = std::max(most, load[x]);
      }
      start = 8 * most;   // blocks past an XCD's list return at once
    } else {
      G.xcd = 2;
    }
  }
  double gfl = 0.0, gby = 0.0;
  for (int i = 0; i < n; ++i) {
    gfl += 2.0 * probs[i].N1 * probs[i].N2 * probs[i].K;
    gby += ((double)probs[i].K * probs[i].N1 + (double)probs[i].K * probs[i].N2) * (dtype == EMO_BF16 ? 2.0 : 4.0) + 4.0 * probs[i].N1 * probs[i].N2;
  }
  EmoTimerScope timer_(EMO_TIMER_GEMM_TN, s, gfl, gby);
  emo_timer_begin(EMO_TIMER_TN_GROUPED, s, gfl, gby);
  if (dtype == EMO_BF16) {
    if (bt == 128 && kb == 1) {
      if (g_tr_read) gemm_tn_grouped_kernel<bf16, true, 1, 128><<<start, 256, 0, s>>>(G);
      else gemm_tn_grouped_kernel<bf16, false, 1, 128><<<start, 256, 0, s>>>(G);
    } else if (bt == 128) {
      if (g_tr_read) gemm_tn_grouped_kernel<bf16, true, 2, 128><<<start, 256, 0, s>>>(G);
      else gemm_tn_grouped_kernel<bf16, false, 2, 128><<<start, 256, 0, s>>>(G);
    } else {
      if (g_tr_read) gemm_tn_grouped_kernel<bf16, true, 2, 64><<<start, 256, 0, s>>>(G);
      else gemm_tn_grouped_kernel<bf16, false, 2, 64><<<start, 256, 0, s>>>(G);
    }
  } else if (split) {
    if (bt == 128) gemm_tn_grouped_kernel<float, true, 1, 128, true><<<start, 256, 0, s>>>(G);
    else gemm_tn_grouped_kernel<float, true, 1, 64, true><<<start, 256, 0, s>>>(G);
  } else if (emo_is_f32(dtype)) {
    if (bt == 128) gemm_tn_grouped_kernel<float, true, 1, 128><<<start, 256, 0, s>>>(G);
    else gemm_tn_grouped_kernel<float, true, 1, 64><<<start, 256, 0, s>>>(G);
  } else {
    emo_set_error("bad dtype %d", dtype);
    return 1;
  }
  emo_timer_end(EMO_TIMER_TN_GROUPED, s);
  EMO_LAUNCH_CHECK();
  return 0;
}

// Conv2d(C->C, k3, s2) over channels-last y1[B,T1,F1,C] as an implicit GEMM:
//   y2[(b,t2,f2), n] = relu(bias[n] + sum_{kh,kw,c} y1[b,2t2+kh,2f2+kw,c] * W[n,(kh,kw,c)])
extern "C" int emoasr_conv2_fwd(int dtype, int B, int T1, int F1, int C, const void* y1, const void* w,
                                void* y2, const emoasr_epilogue_t* ep, void* stream) {
  SplitScope split_scope(dtype);
  EMO_CHECK(T1 >= 3 && F1 >= 3, "conv2: input too small (T1=%d F1=%d)", T1, F1);
  const int T2 = (T1 - 3) / 2 + 1, F2 = (F1 - 3) / 2 + 1;
  EMO_CHECK(C % 32 == 0, "conv2: C must be a multiple of 32");
  // bias (+ ReLU) only, bf16, C % 256 == 0: the large-tile kernel (gemm_big.hip)
  if (dtype == EMO_BF16 && C % 256 == 0 && emo_conv_big_enabled() && ep->alpha == 1.f && !ep->residual && !ep->dact_pre &&
      !ep->pre_out && ep->drop_p == 0.f && !ep->out_f32 && (ep->act == EMO_ACT_NONE || ep->act == EMO_ACT_RELU))
    return emo_conv2_fwd_big(B, T1, F1, C, y1, w, y2, ep->bias, ep->act == EMO_ACT_RELU, (hipStream_t)stream);
  NtArgs a{};
  a.M = B * T2 * F2; a.N = C; a.K = 9 * C; a.A = y1; a.lda = 0; a.B = w; a.ldb = 9 * C; a.C = y2; a.ldc = C;
  a.ep = *ep;
  a.cg = ConvGeom{T1, F1, T2, F2, C};
  EMO_DISPATCH(dtype, return (launch_nt<T, 1>(a, (hipStream_t)stream)));
}

// dy1[b,t1,f1,c] = relu'(y1[b,t1,f1,c]) * sum_{kh,kw,n} dy2[b,(t1-kh)/2,(f1-kw)/2,n] * W[n,(kh,kw,c)]
// (terms with odd / out-of-range source indices vanish): four implicit GEMMs, one per parity class of
// (t1, f1), with K = 4C / 2C / 2C / C -- see DgradGeom.  w = the forward's [C, 9C] weight layout.
extern "C" int emoasr_conv2_dgrad(int dtype, int B, int T1, int F1, int C, const void* dy2, const void* w,
                                  const void* y1, void* dy1, void* stream) {
  SplitScope split_scope(dtype);
  EMO_CHECK(T1 >= 3 && F1 >= 3, "conv2_dgrad: input too small (T1=%d F1=%d)", T1, F1);
  EMO_CHECK(C % 64 == 0, "conv2_dgrad: C must be a multiple of 64");
  const int T2 = (T1 - 3) / 2 + 1, F2 = (F1 - 3) / 2 + 1;
  for (int pt = 0; pt < 2; ++pt)
    for (int pf = 0; pf < 2; ++pf) {
      NtArgs a{};
      DgradGeom& g = a.dg;
      g.T1 = T1; g.F1 = F1; g.T2 = T2; g.F2 = F2; g.C = C; g.pt = pt; g.pf = pf;
      g.nI = (T1 - pt + 1) / 2; g.nJ = (F1 - pf + 1) / 2;
      g.ntap = 0;
      for (int kh = pt; kh < 3; kh += 2)
        for (int kw = pf; kw < 3; kw += 2) {
          g.dh[g.ntap] = kh / 2; g.dw[g.ntap] = kw / 2; g.wtap[g.ntap] = (kh * 3 + kw) * C;
          ++g.ntap;
        }
      if (g.nI <= 0 || g.nJ <= 0) continue;
      a.M = B * g.nI * g.nJ; a.N = C; a.K = g.ntap * C;
      a.A = dy2; a.lda = 0; a.B = w; a.ldb = 9 * C; a.C = dy1; a.ldc = C;
      a.ep.alpha = 1.f; a.ep.dact_pre = y1; a.ep.dact = EMO_ACT_RELU; a.ep.res_scale = 1.f;
      int rc = 1;
      EMO_DISPATCH(dtype, rc = (is_f32<T> && t_f32_split ? launch_nt_<float, 2, true, true, true>(a, (hipStream_t)stream)
                                : g_tr_read ? launch_nt_<T, 2, true, true>(a, (hipStream_t)stream)
                                            : launch_nt_<T, 2, true, false>(a, (hipStream_t)stream)));
      if (rc) return rc;
    }
  return 0;
}

// dW[n, (kh,kw,c)] (+)= sum_{(b,t2,f2)} dy2[(b,t2,f2), n] * y1[b,2t2+kh,2f2+kw,c]
extern "C" int emoasr_conv2_wgrad(int dtype, int B, int T1, int F1, int C, const void* dy2, const void* y1,
                                  float* dw, float* dbias, int accumulate, void* stream) {
  SplitScope split_scope(dtype);
  const int T2 = (T1 - 3) / 2 + 1, F2 = (F1 - 3) / 2 + 1;
  EMO_CHECK(T1 >= 3 && F1 >= 3, "conv2_wgrad: input too small");
  EMO_CHECK(C % 128 == 0, "conv2_wgrad: C must be a multiple of 128");
  if (!accumulate) {
    hipMemsetAsync(dw, 0, sizeof(float) * (size_t)C * 9 * C, (hipStream_t)stream);
    if (dbias) hipMemsetAsync(dbias, 0, sizeof(float) * C, (hipStream_t)stream);
  }
  TnArgs a{};
  a.N1 = C; a.N2 = 9 * C; a.K = B * T2 * F2; a.A = dy2; a.lda = C; a.B = y1; a.ldb = 0; a.C = dw;
  a.ldc = 9 * C; a.alpha = 1.f; a.colsum = dbias; a.colsum_scale = 1.f;
  a.cg = ConvGeom{T1, F1, T2, F2, C};
  EMO_DISPATCH(dtype, return (launch_tn<T, 1>(a, (hipStream_t)stream)));
}
