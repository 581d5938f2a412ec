// Large-tile bf16 MFMA GEMM for the one compute-bound product family of the encoder: the Conv2d(256 -> 256, k3, s2)
// of the subsampling front-end (asr/modeling/encoders/conv.py:9-19) -- forward, data gradient and weight gradient are
// each ~157 GFLOP per 27 k-frame batch (M = B*T2*F2 ~ 133 k rows, K = 9*256, N = 256), 13 % of the step when run on
// the 128x64-tile kernel of gemm.hip (LDS-read-bound at ~20 % of the MFMA peak).
//
//   big_nt : C[M,N] = epilogue(A[M,K] . B[N,K]^T), A plain / gathered through the conv geometry (forward) / gathered
//            per output-parity class (data gradient; all four classes in one launch)
//
// Structure (one workgroup per CU):
//   * 512 threads = 8 waves as 2 (M) x 4 (N); block tile BM x 256 with BM = 256 / 192 / 128 chosen by the host so that
//     the tile count fills whole rounds of the 256 CUs; wave tile (BM/2) x 64 of 16x16x32 MFMAs (128 accumulator
//     registers at BM = 256).  Per k-tile and wave: (BM/2 + 64) * 128 B of LDS reads for BM/2 * 64 * 64 * 2 flop.
//   * k-tiles of 64; LDS rings of three A stages (BM x 128 B each) and two B stages (256 x 128 B): 160 KB at BM = 256,
//     filled by LDS-DMA (buffer_load_dwordx4 ... lds: no staging registers, no ds_write pass): each wave-instruction
//     lands 8 rows x 128 B.
//     The image is lane-linear, so the bank swizzle is applied on the SOURCE side: the 16-byte chunk a lane fetches is
//     chunk ^ ((row >> 1) & 7), and fragment reads (ds_read_b128, 16 rows x 16 B per 16-lane group) use the same XOR:
//     conflict-free for the 16x16x32 operand shape.
//   * one raw s_barrier per k-tile and a counted vmcnt wait (the newest A tile stays in flight across it); DMA pieces
//     are issued between the MFMA groups of the running tile, fragment reads one 8-MFMA step ahead of their use.
//   * measured at M = 130 663 (MI355X, HIP-graph timed): forward 141 us = 1.09 PFLOP/s (the 128x64 kernel: 290-317 us);
//     timing ablations of that launch: MFMAs alone 91-95 us (1.65 PFLOP/s: the ceiling of 2 rounds of 256-row tiles at the
//     sustained matrix clock), + fragment reads 104, + barrier 109-112, + DMA 141-146.
//   * out-of-range rows / taps use the buffer descriptor's bounds check (offset 0xFFFFFFFF -> zeros land in LDS).
//   * operands are multiplied as D^T = B . A^T, so a lane ends up with 4 CONSECUTIVE output columns of one row; the
//     epilogue packs them to 8 bytes, transposes 16-row slabs through wave-private LDS and stores full 128-byte rows.
#include <algorithm>
#include "mma.h"
#include "../../include/emoasr_hip.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_void_ptr;

struct BigConv {   // forward gather: see ConvGeom in gemm.hip
  int T1, F1, T2, F2, C;
};
struct BigDgrad {  // data gradient, one entry per output-parity class (pt, pf)
  int T1, F1, T2, F2, C;
  int ncls;
  int tile0[5];            // first tile of each class
  int pt[4], pf[4], nI[4], nJ[4], ntap[4], M[4];
  int dh[4][4], dw[4][4], wtap[4][4];
};
// Transducer joint head without the [B,T,U,V] logits (rnn_transducer.py:101-115,147-156): epilogue modes of the plain product
// z = h . W^T + bias over rows n = lattice cells (b, t, u).
//   mode 1 (forward): nothing is stored but, per row and 64-column chunk, (max, sum exp(z - max)) -> part[n, chunk, 2], and the two
//                     logits the lattice needs: zb[n] = z[n, blank], zy[n] = z[n, label(b, u)];
//   mode 2 (gradient, rows of one chunk): C[n, v] = exp(z - lse[n]) * occ[n] - [v == blank] gb[n] - [v == y[n]] gy[n], the four
//                     row constants (already scaled) in coef[n, 4], the label column in ycol[n];
//   mode 3 (CTC head): the product is stored AND the partials of mode 1 are produced (of the values as rounded to bf16): the
//                     vocabulary projection and the soft-max denominators in one pass over the logits.
struct BigRnnt {
  int mode, Tn, U, Lmax, blank, nchunk;
  long row0;               // cell index of row 0 of this launch
  long part_rows, part_row0;   // partial table [chunk][part_rows][2]: this launch's row 0 is table row part_row0
  const int* labels; const int* ylens;
  float* part; float* zb; float* zy;
  const float* coef; const int* ycol;
};
struct BigArgs {
  int M, N, K;
  const void* A; long lda;
  const void* B; long ldb;
  void* C; long ldc;
  const float* bias;
  int relu;
  emoasr_epilogue_t ep;  // AMODE 0 only: the general epilogue of emoasr_gemm_nt (no residual / dact_pre / f32 output)
  BigRnnt rn;            // AMODE 0 only: transducer-head epilogues (mode 0 = off)
  const void* dmask;  // data gradient only, optional: multiply by (dmask[same offset as C] > 0)  (ReLU backward)
  int tiles_m, tiles_n;
  int korder;  // gathered modes: 1 = channel chunk outermost, taps innermost; 0 = tap-major
  BigConv cg;
  BigDgrad dg;
};

__device__ __forceinline__ int xcd_remap_big(int pid, int nblk) {
  const int per = nblk / 8, rem = nblk - per * 8;
  const int x = pid % 8, slot = pid / 8;
  return x * per + min(x, rem) + slot;
}

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* lds, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_ptr)lds, 16, voff, soff, 0, 0);
}

// exchange inside groups of 8 lanes on the VALU (DPP), for reductions: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror
// (lane i <-> 7 - i of its half-row: after the two quad steps every lane of a quad holds the quad's result, so the mirror pairs the
// two quads' results).  __shfl_xor compiles to ds_bpermute_b32 -- an LDS-pipe round trip per step.
template <int CTRL> __device__ __forceinline__ float dpp8(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float max8(float x) {
  x = fmaxf(x, dpp8<0xB1>(x)); x = fmaxf(x, dpp8<0x4E>(x)); return fmaxf(x, dpp8<0x141>(x));
}
__device__ __forceinline__ float sum8(float x) {
  x += dpp8<0xB1>(x); x += dpp8<0x4E>(x); return x + dpp8<0x141>(x);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  static_assert(N == 8 || N == 6 || N == 4 || N == 3 || N == 2, "wait_vmcnt: unexpected piece count");
}

// AMODE 0: plain row-major A.  1: Conv2d forward gather.  2: Conv2d data gradient (parity classes).
// NW = waves per workgroup: 8 (2 x 4 waves of (BM/2) x 64) or, round 4, 4 (2 x 2 waves of (BM/2) x 128 -- one wave per SIMD with
// 256 accumulator registers at BM = 256: per k-tile and wave (BM/2 + 128) * 128 B of LDS reads for TWICE the flops of the 8-wave
// layout, whose fragment reads alone need 85 % of the LDS bandwidth at the MFMA peak).
// MODE (AMODE 0 only): the epilogue variant -- 0 general, 1 / 2 / 3 the transducer / CTC head modes of BigRnnt -- as a TEMPLATE
// parameter: with the mode a run-time field every instantiation carried all four fully unrolled epilogues (20 k instructions
// for TMW = 4; a wave runs the epilogue once per tile, straight through: SQ_WAIT_INST_ANY was 43 % of the wave cycles of the
// head-gradient launch, i.e. instruction fetch).
template <int TMW, int AMODE, int NW, int MODE>
__global__ __launch_bounds__(NW * 64) void big_nt_kernel(const BigArgs g) {
  static_assert(AMODE == 0 || MODE == 0, "epilogue modes belong to the plain product");
  constexpr int BM = TMW * 32, BN = 256;
  constexpr int NJ = BN / (NW / 2) / 16;           // 16-column groups per wave: 4 (64 columns) or 8 (128)
  constexpr int WCOLS = NJ * 16;
  constexpr int LPR = WCOLS / 8, HH = 16 / (64 / LPR);   // epilogue: lanes per slab row, passes of 64 lanes over a 16-row slab
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  constexpr int A_PCS = BM / (NW * 8), B_PCS = BN / (NW * 8);  // 1-KiB pieces (8 rows x 128 B) per wave and tile
  constexpr int EP_LD = WCOLS * 2 + 16;            // bytes per row of the wave-private transpose slab (bf16 row + 16)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / (NW / 2), wc = wave % (NW / 2);
  // XCD-contiguous tile ranges (neighbouring row tiles share the convolution's halo rows in one L2); not for the data
  // gradient, whose tiles are ordered heaviest class first for the dispatcher
  const int bid = AMODE == 2 ? (int)blockIdx.x : xcd_remap_big(blockIdx.x, gridDim.x);

  // ---- which tile -----------------------------------------------------------------------------------
  int cls = 0, tm, tn;
  int M = g.M;
  if constexpr (AMODE == 2) {
    const int t = bid / g.tiles_n;
    tn = bid - t * g.tiles_n;
    while (cls + 1 < g.dg.ncls && t >= g.dg.tile0[cls + 1]) ++cls;
    tm = t - g.dg.tile0[cls];
    M = g.dg.M[cls];
  } else if (AMODE == 0 && g.tiles_n > 1) {
    // wide products (the vocabulary projection: 40 column tiles, W = 5 MB > one XCD's 4 MB of L2): groups of 8 row tiles,
    // column-major inside a group -- the 32 tiles an XCD runs at once are 8 row bands x 4 weight tiles, and a weight tile
    // serves 8 consecutive workgroups.  (Row-major order read the whole weight once per row band: 971 MB for 23 MB of operands.)
    constexpr int GM = 8;
    const int per = GM * g.tiles_n, grp = bid / per, first = grp * GM;
    const int gsz = min(GM, g.tiles_m - first), local = bid - grp * per;
    tn = local / gsz;
    tm = first + local - tn * gsz;
  } else {
    tm = bid / g.tiles_n;
    tn = bid - tm * g.tiles_n;
  }
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- DMA source offsets (bytes; per lane one row of each of its pieces) ------------------------------
  const int sub = lane >> 3, pc = lane & 7;
  unsigned a_off[A_PCS], b_off[B_PCS];
  unsigned a_tapmask = 0;  // AMODE 2: 4 bits per piece, bit (dh*2 + dw) = that shifted source row exists
  const char* a_base = static_cast<const char*>(g.A);
#pragma unroll
  for (int i = 0; i < A_PCS; ++i) {
    const int row = (wave + NW * i) * 8 + sub;
    const unsigned ch = (unsigned)(pc ^ ((row >> 1) & 7)) * 16u;
    const int grow = m0 + row;
    const bool ok = grow < M;
    const int r = ok ? grow : 0;
    if constexpr (AMODE == 1) {
      const int per_b = g.cg.T2 * g.cg.F2;
      const int b = r / per_b, q = r - b * per_b;
      const int t2 = q / g.cg.F2, f2 = q - t2 * g.cg.F2;
      a_off[i] = ok ? (unsigned)(((((long)b * g.cg.T1 + 2 * t2) * g.cg.F1 + 2 * f2) * g.cg.C) * 2) + ch : EMO_OOB;
    } else if constexpr (AMODE == 2) {
      const int per_b = g.dg.nI[cls] * g.dg.nJ[cls];
      const int b = r / per_b, q = r - b * per_b;
      const int ii = q / g.dg.nJ[cls], jj = q - ii * g.dg.nJ[cls];
      // offsets are relative to dy2 - (F2 + 1) * C elements, so that the tap shifts (dh, dw in {0, 1}) are non-negative
      a_off[i] = (unsigned)(((((long)b * g.dg.T2 + ii) * g.dg.F2 + jj) * g.dg.C) * 2) + ch;
      unsigned msk = 0;
#pragma unroll
      for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int dw = 0; dw < 2; ++dw) {
          const int si = ii - dh, sj = jj - dw;
          if (ok && si >= 0 && si < g.dg.T2 && sj >= 0 && sj < g.dg.F2) msk |= 1u << (dh * 2 + dw);
        }
      a_tapmask |= msk << (4 * i);
    } else {
      a_off[i] = ok ? (unsigned)((long)r * g.lda * 2) + ch : EMO_OOB;
    }
  }
  if constexpr (AMODE == 2) a_base -= (long)(g.dg.F2 + 1) * g.dg.C * 2;
#pragma unroll
  for (int i = 0; i < B_PCS; ++i) {
    const int row = (wave + NW * i) * 8 + sub;
    const unsigned ch = (unsigned)(pc ^ ((row >> 1) & 7)) * 16u;
    const int n = n0 + row;
    b_off[i] = n < g.N ? (unsigned)((long)n * g.ldb * 2) + ch : EMO_OOB;
  }
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(a_base), rsB = make_rsrc(g.B);

  const int cpt = (AMODE == 0) ? 1 : (AMODE == 1 ? g.cg.C : g.dg.C) / 64;  // k-tiles per tap
  const int nk = (AMODE == 2) ? g.dg.ntap[cls] * cpt : g.K / 64;

  // Staging rings.  The B operand (the weight: 1.2 MB, re-read by every tile) always hits the L2; the A operand streams
  // from HBM / Infinity Cache with ~2 us of latency under load, more than one k-tile of MFMA work (~1.5 us).  With two
  // symmetric stages the wait for the LAST piece of the next tile was exposed in every k-tile (measured: MFMA + fragment
  // reads + barrier alone 109 us, DMA alone 63 us, together 146 us).  So the 160 KB of LDS are split unevenly: B two stages
  // (tile kt + 1 in flight), A THREE stages (tiles kt + 1 and kt + 2 in flight).  Within an iteration the B pieces are
  // issued before the A pieces, so that the counted wait at the top of the next iteration, vmcnt(A_PCS), retires
  // everything but the newest A tile.
  // k order of the gathered modes: channel chunk outermost, taps innermost (neighbouring taps read nearly the same input
  // rows, so their re-reads follow the first touch by one to three k-tiles and hit the XCD's L2: fabric reads of the
  // forward 621 -> 337 MB).
  auto tile_offsets = [&](int kt, unsigned& sA, unsigned& sB, int& tapbit) {
    tapbit = 0;
    if constexpr (AMODE == 1) {
      const int kc = g.korder ? kt / 9 : kt % cpt, tap = g.korder ? kt - kc * 9 : kt / cpt;
      const int kh = tap / 3, kw = tap - kh * 3;
      sA = (unsigned)(((kh * g.cg.F1 + kw) * g.cg.C + kc * 64) * 2);
      sB = (unsigned)((tap * g.cg.C + kc * 64) * 2);
    } else if constexpr (AMODE == 2) {
      const int ntap = g.dg.ntap[cls];
      const int kc = g.korder ? kt / ntap : kt % cpt, tap = g.korder ? kt - kc * ntap : kt / cpt;
      const int dh = g.dg.dh[cls][tap], dw = g.dg.dw[cls][tap];
      sA = (unsigned)((((1 - dh) * g.dg.F2 + (1 - dw)) * g.dg.C + kc * 64) * 2);
      sB = (unsigned)((g.dg.wtap[cls][tap] + kc * 64) * 2);
      tapbit = dh * 2 + dw;
    } else {
      sA = sB = (unsigned)kt * 128u;
    }
  };
  constexpr int A_RING = 3 * A_BYTES;  // B ring starts here
  unsigned ia_s = 0, ib_s = 0;         // scalar source offsets of the tiles being issued
  int ia_tapbit = 0;
  char* ia_dst = smem;                 // this wave's first piece of the A / B stage being filled
  char* ib_dst = smem;
  auto setup_a = [&](int kt) {
    unsigned dummy; 
    tile_offsets(kt, ia_s, dummy, ia_tapbit);
    ia_dst = smem + (kt % 3) * A_BYTES + wave * 1024;
  };
  auto setup_b = [&](int kt) {
    unsigned dummy;
    int tb;
    tile_offsets(kt, dummy, ib_s, tb);
    ib_dst = smem + A_RING + (kt & 1) * B_BYTES + wave * 1024;
  };
  auto issue_a = [&](const int p) __attribute__((always_inline)) {
    unsigned v = a_off[p];
    if constexpr (AMODE == 2) v = ((a_tapmask >> (4 * p + ia_tapbit)) & 1u) ? v : EMO_OOB;
    dma16(rsA, ia_dst + p * (NW * 1024), v, ia_s);
  };
  auto issue_b = [&](const int p) __attribute__((always_inline)) { dma16(rsB, ib_dst + p * (NW * 1024), b_off[p], ib_s); };
  constexpr int N_PCS = A_PCS + B_PCS;

  // ---- fragment read offsets -----------------------------------------------------------------------
  // lane -> row (lane & 15) of a 16-row group, logical chunk ks*4 + (lane >> 4), swizzled by ((row >> 1) & 7);
  // group bases are multiples of 16 rows, so the XOR term depends on the lane only
  const int frow = lane & 15;
  const unsigned fsw = (unsigned)((lane >> 1) & 7);
  unsigned fch[2];
  fch[0] = (((unsigned)(lane >> 4)) ^ fsw) * 16u;
  fch[1] = (((unsigned)(4 + (lane >> 4))) ^ fsw) * 16u;
  const unsigned a_frag0 = (unsigned)((wr * (BM / 2) + frow) * 128);
  const unsigned b_frag0 = (unsigned)(A_RING + (wc * WCOLS + frow) * 128);

  f32x4 acc[TMW][NJ];
#pragma unroll
  for (int i = 0; i < TMW; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // prologue: A(0), B(0), then A(1) -- the one group that may stay in flight across the first wait
  setup_a(0);
#pragma unroll
  for (int p = 0; p < A_PCS; ++p) issue_a(p);
  setup_b(0);
#pragma unroll
  for (int p = 0; p < B_PCS; ++p) issue_b(p);
  if (nk > 1) {
    setup_a(1);
#pragma unroll
    for (int p = 0; p < A_PCS; ++p) issue_a(p);
  }
  // MFMA groups (ks, i) per k-tile: 2 * TMW; piece p (B pieces first) goes out after group p * (2 * TMW - 2) / N_PCS
  constexpr int GROUPS = 2 * TMW;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      wait_vmcnt<A_PCS>();
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const bool more_b = kt + 1 < nk, more_a = kt + 2 < nk;
    if (more_b) setup_b(kt + 1);
    if (more_a) setup_a(kt + 2);
    const char* sta = smem + (kt % 3) * A_BYTES;
    const char* stb = smem + (kt & 1) * B_BYTES;
    // Fragment reads run one step ahead of the MFMAs that use them: a step = one pair of 16-row A groups against the
    // four B groups of one 32-deep k slice (8 MFMAs); while it runs, the next pair (and, towards the end of slice 0,
    // slice 1's B groups) is read into the other register set, so no MFMA waits on a read issued just before it.
    constexpr int PAIRS = TMW / 2, STEPS = 2 * PAIRS;
    bf16x8 bq[2][NJ], aq[2][2];
    auto ld_b = [&](int ks, int j) { return *reinterpret_cast<const bf16x8*>(stb + b_frag0 + j * 2048 + fch[ks]); };
    auto ld_a = [&](int ks, int i) { return *reinterpret_cast<const bf16x8*>(sta + a_frag0 + i * 2048 + fch[ks]); };
#pragma unroll
    for (int j = 0; j < NJ; ++j) bq[0][j] = ld_b(0, j);
    aq[0][0] = ld_a(0, 0);
    aq[0][1] = ld_a(0, 1);
#pragma unroll
    for (int sidx = 0; sidx < STEPS; ++sidx) {
      const int ks = sidx / PAIRS, pr = sidx % PAIRS;
      if (sidx + 1 < STEPS) {
        const int ks1 = (sidx + 1) / PAIRS, pr1 = (sidx + 1) % PAIRS;
        aq[(sidx + 1) & 1][0] = ld_a(ks1, 2 * pr1);
        aq[(sidx + 1) & 1][1] = ld_a(ks1, 2 * pr1 + 1);
      }
      if (PAIRS >= 2 && sidx == PAIRS - 2) {
#pragma unroll
        for (int j = 0; j < NJ / 2; ++j) bq[1][j] = ld_b(1, j);
      }
      if (PAIRS >= 2 && sidx == PAIRS - 1) {
#pragma unroll
        for (int j = NJ / 2; j < NJ; ++j) bq[1][j] = ld_b(1, j);
      }
      __builtin_amdgcn_sched_barrier(0);  // the scheduler otherwise sinks these reads to just before their first use
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[2 * pr + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[ks][j], aq[sidx & 1][i], acc[2 * pr + i][j], 0, 0, 0);
        const int grp = ks * TMW + 2 * pr + i;
#pragma unroll
        for (int p = 0; p < N_PCS; ++p)
          if (grp == (p * (GROUPS - 2)) / N_PCS) {
            if (p < B_PCS) { if (more_b) issue_b(p); }
            else { if (more_a) issue_a(p - B_PCS); }
          }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue -------------------------------------------------------------------------------------
  // acc[i][j][r] = C[m = group i, row lane & 15][n = group j, col 4 * (lane >> 4) + r]
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // every wave is done with the staging buffers
  const int ncol0 = n0 + wc * WCOLS;
  if constexpr (AMODE == 0) {
    // general epilogue, in the order and precision of gemm_nt_kernel's: alpha, bias, pre_out, activation, dropout (mask
    // indexed by row * N + col).  16-row slabs go through wave-private LDS as f32 (row stride 68 floats), then every lane
    // owns 8 consecutive columns of a row.
    constexpr int FLD = WCOLS + 4;
    float* fslab = reinterpret_cast<float*>(smem) + wave * (16 * FLD);
    const emoasr_epilogue_t& ep = g.ep;
    bf16* Cp = static_cast<bf16*>(g.C);
    bf16* pre_out = static_cast<bf16*>(ep.pre_out);
    if constexpr (MODE >= 1 && MODE <= 3) {
      const BigRnnt& rn = g.rn;
      const int chunk0 = ncol0 >> 6;
      // mode 2: the row constants of the tile's BM rows go through LDS once (behind the wave-private slabs) -- read per row from
      // global memory inside the slab loop they were 2 * TMW dependent round trips per wave
      float* cfs = reinterpret_cast<float*>(smem + 40960);          // [BM][4]
      int* ycs = reinterpret_cast<int*>(smem + 40960 + BM * 16);    // [BM]
      if constexpr (MODE == 2) {
        if (tid < BM) {
          const int grow = m0 + tid;
          const bool ok = grow < M;
          *reinterpret_cast<f32x4*>(cfs + tid * 4) = ok ? *reinterpret_cast<const f32x4*>(rn.coef + (long)grow * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
          ycs[tid] = ok ? rn.ycol[grow] : -1;
        }
        __syncthreads();
      }
      if constexpr (MODE == 1) {
        // the label column of the tile's rows (emoasr_rnnt_ycol: -1 = none) through LDS: formed per row inside the slab loop it
        // was two integer divisions and two DEPENDENT global loads (ylens[b], labels[b, u]) per pass
        if (tid < BM) ycs[tid] = (m0 + tid < M) ? rn.ycol[m0 + tid] : -1;
        __syncthreads();
      }
#pragma unroll
      for (int i = 0; i < TMW; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) *reinterpret_cast<f32x4*>(fslab + frow * FLD + j * 16 + 4 * (lane >> 4)) = acc[i][j];
#pragma unroll
        for (int hh = 0; hh < HH; ++hh) {
          const int id = lane + 64 * hh, row = id / LPR, cc = id % LPR;
          const int chunk = chunk0 + (cc >> 3);   // (eight lanes = one 64-column chunk of the row)
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(fslab + row * FLD + cc * 8);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(fslab + row * FLD + cc * 8 + 4);
          const int grow = m0 + wr * (BM / 2) + i * 16 + row;
          const int col = ncol0 + cc * 8;
          const bool rok = grow < M, cok = col < g.N;   // (N % 8 == 0: a lane's eight columns are all inside or all outside)
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = v0[e]; v[4 + e] = v1[e]; }
          if (ep.bias && cok) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(ep.bias + col);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(ep.bias + col + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
          }
          if (MODE == 3 && cok) {
            // (the partials must describe the logits AS STORED: round to the output type first, so that exp(z - lse) of the stored
            // row sums to one exactly as after a separate row pass)
            if (rok) store8<bf16>(Cp + (long)grow * g.ldc + col, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)(bf16)v[e];
          }
          if constexpr (MODE == 1 || MODE == 3) {
            // the eight lanes of a row (lane bits 0..2) hold its 64 columns of this wave's chunk: the chunk's maximum first (three
            // DPP steps), then the sum of exp(z - max) (three more) -- a butterfly over (max, sum) pairs cost two ds_bpermute and two
            // more exponentials per step
            float m = -INFINITY, sm = 0.f;
            if (cok) {
#pragma unroll
              for (int e = 0; e < 8; ++e) m = fmaxf(m, v[e]);
            }
            m = max8(m);
            if (cok) {
#pragma unroll
              for (int e = 0; e < 8; ++e) sm += __expf(v[e] - m);
            }
            sm = sum8(sm);
            if (rok) {
              if ((cc & 7) == 0 && chunk < rn.nchunk) {   // (the last column tile may reach past ceil(N / 64) chunks)
                // chunk-major table: the eight rows of this pass land in 64 consecutive bytes, and the fold reads it coalesced
                // (row-major, 8-byte pieces 8 * nchunk bytes apart: the fold of the CTC head's 35 k x 157 table took 103 us)
                *reinterpret_cast<float2*>(rn.part + ((long)chunk * rn.part_rows + rn.part_row0 + grow) * 2) = float2{m, sm};
              }
              if (cok && MODE == 1) {
                const int kb = rn.blank - col, ky = ycs[grow - m0] - col;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  if (kb == e) rn.zb[grow] = v[e];
                  if (ky == e) rn.zy[grow] = v[e];
                }
              }
            }
          } else if (rok && cok) {
            const int lrow = grow - m0;
            const f32x4 cf = *reinterpret_cast<const f32x4*>(cfs + lrow * 4);   // lse, occ, gb, gy (scaled)
            const int kb = rn.blank - col, ky = ycs[lrow] - col;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float gq = __expf(v[e] - cf[0]) * cf[1];
              if (kb == e) gq -= cf[2];
              if (ky == e) gq -= cf[3];
              v[e] = gq;
            }
            store8<bf16>(Cp + (long)grow * g.ldc + col, v);
          }
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) *reinterpret_cast<f32x4*>(fslab + frow * FLD + j * 16 + 4 * (lane >> 4)) = acc[i][j];
#pragma unroll
      for (int hh = 0; hh < HH; ++hh) {
        const int id = lane + 64 * hh, row = id / LPR, cc = id % LPR;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(fslab + row * FLD + cc * 8);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(fslab + row * FLD + cc * 8 + 4);
        const int grow = m0 + wr * (BM / 2) + i * 16 + row;
        const int col = ncol0 + cc * 8;
        if (grow < M && col < g.N) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = v0[e]; v[4 + e] = v1[e]; }
          if (ep.bias) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(ep.bias + col);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(ep.bias + col + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = ep.alpha * v[e] + b0[e]; v[4 + e] = ep.alpha * v[4 + e] + b1[e]; }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= ep.alpha;
          }
          const long off = (long)grow * g.ldc + col;
          if constexpr (MODE == 5) {
            // the first feed-forward product's epilogue alone: Swish + the saved gradient factor + dropout (EMO_ACT_SAVE_DACT)
            float dd[8], mm[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float sg = sigmoidf_(v[e]), a = v[e] * sg;
              dd[e] = sg + a * (1.f - sg);
              v[e] = a;
            }
            dropout_mult8(ep.seed, (uint64_t)grow * (uint64_t)g.N + col, ep.drop_p, mm);
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[e] *= mm[e]; dd[e] *= mm[e]; }
            if (pre_out) store8<bf16>(pre_out + off, dd);
          } else if constexpr (MODE == 4) {
            // LEAN general epilogue (no activation, no saved tensor; act' in {the saved factor, 1 - tanh^2 of the saved output}):
            // what the long reductions routed here need (front-end Linear, vocabulary head's data gradient, the joint's data
            // gradient, residual outputs).  The full epilogue below inlines the Swish / GELU / tanh / erf ladders of act_vec and
            // dact_vec at each of its 2 * TMW sites: 17 k instructions for TMW = 4.
            if (ep.dact_pre) {
              const Vec16<bf16> pv = load16(static_cast<const bf16*>(ep.dact_pre) + off);
              if (ep.dact == EMO_DACT_MUL) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= pv.get(e);
              } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float t = pv.get(e); v[e] *= 1.f - t * t; }
              }
            }
            dropout_apply8(ep.seed, (uint64_t)grow * (uint64_t)g.N + col, ep.drop_p, v);
          } else if (ep.act & EMO_ACT_SAVE_DACT) {   // (common.h: the saved tensor is act'(pre) * dropout_scale)
            float dd[8], mm[8];
            act_dact_vec<8>(ep.act & 0xFF, v, dd);
            dropout_mult8(ep.seed, (uint64_t)grow * (uint64_t)g.N + col, ep.drop_p, mm);
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[e] *= mm[e]; dd[e] *= mm[e]; }
            if (pre_out) store8<bf16>(pre_out + off, dd);
          } else {
            if (pre_out) store8<bf16>(pre_out + off, v);
            act_vec<8>(ep.act, v);
            if (ep.dact_pre) {   // data gradient through an activation: times act'(saved tensor), same offsets as C
              const Vec16<bf16> pv = load16(static_cast<const bf16*>(ep.dact_pre) + off);
              float d[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) d[e] = pv.get(e);
              dact_vec<8>(ep.dact, d, v);
            }
            dropout_apply8(ep.seed, (uint64_t)grow * (uint64_t)g.N + col, ep.drop_p, v);   // (N % 8 == 0, col % 8 == 0)
          }
          if (ep.residual) {   // x + res_scale * (...); C may alias the residual: every lane reads exactly what it then writes
            const Vec16<bf16> rv = load16(static_cast<const bf16*>(ep.residual) + (long)grow * ep.ldr + col);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = rv.get(e) + ep.res_scale * v[e];
          }
          store8<bf16>(Cp + off, v);
        }
      }
    }
    return;
  }
  char* slab = smem + wave * (16 * EP_LD);
  f32x4 bias4[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    bias4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (g.bias) bias4[j] = *reinterpret_cast<const f32x4*>(g.bias + ncol0 + j * 16 + 4 * (lane >> 4));
  }
  bf16* Cp = static_cast<bf16*>(g.C);
  const bf16* Dm = static_cast<const bf16*>(g.dmask);
  // output offsets of this lane's 2 * TMW row segments; the ReLU-mask loads of all of them go out together, before the
  // transposes (one dependent load per store would cost 2 * TMW memory round trips per tile)
  long offs[TMW][HH];
  bf16x8 dm[TMW][HH];
#pragma unroll
  for (int i = 0; i < TMW; ++i)
#pragma unroll
    for (int hh = 0; hh < HH; ++hh) {
      const int id = lane + 64 * hh, row = id / LPR, cc = id % LPR;
      const int grow = m0 + wr * (BM / 2) + i * 16 + row;
      long off = -1;
      if (grow < M) {
        if constexpr (AMODE == 2) {
          const int per_b = g.dg.nI[cls] * g.dg.nJ[cls];
          const int b = grow / per_b, q = grow - b * per_b;
          const int ii = q / g.dg.nJ[cls], jj = q - ii * g.dg.nJ[cls];
          off = ((((long)b * g.dg.T1 + 2 * ii + g.dg.pt[cls]) * g.dg.F1 + 2 * jj + g.dg.pf[cls]) * g.dg.C) + ncol0 + cc * 8;
        } else {
          off = (long)grow * g.ldc + ncol0 + cc * 8;
        }
      }
      offs[i][hh] = off;
      if (AMODE == 2 && Dm) dm[i][hh] = *reinterpret_cast<const bf16x8*>(Dm + (off >= 0 ? off : 0));
    }
#pragma unroll
  for (int i = 0; i < TMW; ++i) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      bf16x4 h;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[i][j][r] + bias4[j][r];
        if (g.relu) v = fmaxf(v, 0.f);
        h[r] = (bf16)v;
      }
      *reinterpret_cast<bf16x4*>(slab + frow * EP_LD + (j * 16 + 4 * (lane >> 4)) * 2) = h;
    }
#pragma unroll
    for (int hh = 0; hh < HH; ++hh) {
      const int id = lane + 64 * hh, row = id / LPR, cc = id % LPR;
      bf16x8 v = *reinterpret_cast<const bf16x8*>(slab + row * EP_LD + cc * 16);
      if (offs[i][hh] >= 0) {
        if (AMODE == 2 && Dm) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (float)dm[i][hh][e] > 0.f ? v[e] : (bf16)0.f;
        }
        *reinterpret_cast<bf16x8*>(Cp + offs[i][hh]) = v;
      }
    }
  }
}

int g_conv_big = 1;

inline int pick_bm(long M, int n_cu) {
  // the tile height that wastes the fewest CU-rounds: efficiency = M / (rounds * n_cu * BM)
  int best = 256;
  double best_eff = 0.0;
  for (int bm : {256, 192, 128}) {
    const long tiles = (M + bm - 1) / bm;
    const long rounds = (tiles + n_cu - 1) / n_cu;
    // a taller tile amortises the 256 weight rows it stages over more output rows
    const double eff = (double)M / ((double)rounds * n_cu * bm) * (bm == 256 ? 1.0 : bm == 192 ? 0.96 : 0.88);
    if (eff > best_eff) { best_eff = eff; best = bm; }
  }
  return best;
}

int n_cu_cached();
// Plain products (AMODE 0): pick_bm's round-efficiency rule as well.  (For one evidence pass of round 4 short reductions took
// 128-row tiles -- measured 1.5x faster for the head kernels -- but that was the instruction-fetch cost of the epilogue growing
// with TMW while all epilogue variants sat in one kernel; with one lean epilogue per instantiation the taller tiles win again:
// CTC head 509 / 453 / 467 us at 128 / 192 / 256 rows, head-gradient chunk 107 / 111 / 95, joint gradient 92 / 89 / 80.)
inline int pick_bm_plain(long M, int N, int K) {
  (void)N; (void)K;
  return pick_bm(M, n_cu_cached());
}

int n_cu_cached() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipGetDevice(&dev);
    hipDeviceProp_t p;
    n = hipGetDeviceProperties(&p, dev) == hipSuccess ? p.multiProcessorCount : 256;
    if (n <= 0) n = 256;
  }
  return n;
}

int g_big_waves = 8;   // option "big_waves": 8 (2 x 4 waves, 64 columns each) or 4 (2 x 2 waves, 128 columns each)

template <int TMW, int AMODE, int NW, int MODE>
int launch_big_t(const BigArgs& a, int tiles, hipStream_t s) {
  constexpr int bytes = (3 * TMW * 32 + 2 * 256) * 128;  // A ring of three stages + B ring of two
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)big_nt_kernel<TMW, AMODE, NW, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { emo_set_error("hipFuncSetAttribute(%d): %s", bytes, hipGetErrorString(e)); return 1; }
    attr_done = true;
  }
  big_nt_kernel<TMW, AMODE, NW, MODE><<<tiles, NW * 64, bytes, s>>>(a);
  EMO_LAUNCH_CHECK();
  return 0;
}

template <int AMODE, int NW, int MODE>
int launch_big_bm_(const BigArgs& a, int bm, int tiles, hipStream_t s) {
  if (bm == 256) return launch_big_t<8, AMODE, NW, MODE>(a, tiles, s);
  if (bm == 192) return launch_big_t<6, AMODE, NW, MODE>(a, tiles, s);
  return launch_big_t<4, AMODE, NW, MODE>(a, tiles, s);
}

template <int AMODE>
int launch_big_bm(const BigArgs& a, int bm, int tiles, hipStream_t s) {
  if constexpr (AMODE == 0) {
    switch (a.rn.mode) {
      case 1: return launch_big_bm_<0, 8, 1>(a, bm, tiles, s);
      case 2: return launch_big_bm_<0, 8, 2>(a, bm, tiles, s);
      case 3: return launch_big_bm_<0, 8, 3>(a, bm, tiles, s);
      default: break;
    }
    const emoasr_epilogue_t& ep = a.ep;
    const bool lean = (ep.act & 0xFF) == EMO_ACT_NONE && !(ep.act & EMO_ACT_SAVE_DACT) && !ep.pre_out &&
                      (!ep.dact_pre || ep.dact == EMO_DACT_MUL || ep.dact == EMO_DACT_TANH_OUT);
    if (lean && g_big_waves != 4) return launch_big_bm_<0, 8, 4>(a, bm, tiles, s);
    if (ep.act == (EMO_ACT_SWISH | EMO_ACT_SAVE_DACT) && !ep.dact_pre && !ep.residual && g_big_waves != 4)
      return launch_big_bm_<0, 8, 5>(a, bm, tiles, s);
  }
  // (the 4-wave layout -- measured slower on every product, DESIGN.md section 7 -- is kept for the general epilogue only)
  if (g_big_waves == 4) return launch_big_bm_<AMODE, 4, 0>(a, bm, tiles, s);
  return launch_big_bm_<AMODE, 8, 0>(a, bm, tiles, s);
}

int g_big_bm = 0;  // tuning override
int g_big_korder = 1;

}  // namespace

void emo_gemm_set_conv_big(int v) { g_conv_big = v; }
void emo_gemm_set_big_bm(int v) { g_big_bm = v; }
void emo_gemm_set_big_waves(int v) { g_big_waves = v == 4 ? 4 : 8; }
void emo_gemm_set_big_korder(int v) { g_big_korder = v; }
int emo_conv_big_enabled() { return g_conv_big; }

// Measured in the L2 training step (M ~ 7 k rows): with the q/k/v, feed-forward w1 and pointwise-conv-1 products
// (165 - 220 tiles of 128 x 256) on this kernel the step took 9.72 ms against 9.66 ms without -- those launches are
// latency-bound either way -- so only products of at least two full rounds of tiles are taken.  Round 3, stacked rows
// (M ~ 35 k): the same three products (550 - 1100 tiles) run FASTER on the 64 x 64 kernel of gemm.hip (step 31.44 -> 30.96 ms
// with the threshold at 2000 tiles, 31.17 with this kernel off for plain products): what stays here is the vocabulary
// projection (N = 10 000: 2 280 tiles at 7 k rows, 11 000 at 35 k) and the Conv2d products.
int g_big_min_tiles = 2000;

// Does the large-tile kernel take this emoasr_gemm_nt call?  (bf16 product, full 256-column tiles, 64-deep k-tiles, an
// epilogue without f32 output, and enough 128-row tiles to occupy most CUs.)
// Round 4: ... and the long reductions onto ONE 256-column tile (N = 256, K >= 512, stacked row counts): the second feed-forward
// product (K = 1024, residual epilogue), the front-end Linear (K = 4864).  A 192-row tile reads every A row once where the 64 x 64
// grid reads it four times; tools/big_n256_probe.py at 35 145 rows: 38.0 -> 29.5 us (K = 1024), 30.9 -> 24.7 (768),
// 184.7 -> 108.3 (4864), 370.7 -> 223.8 (10 048), bit-identical results (vendor BLAS: 23.6 / 20.6 / 87.1 / 191.2).
// Option "big_n256".
int g_big_n256 = 2;
bool emo_gemm_nt_big_wants(int M, int N, int K, long lda, long ldb, long ldc, const emoasr_epilogue_t& ep) {
  if (!g_conv_big || N % 8 != 0 || N < 256 || K % 64 != 0 || lda % 8 != 0 || ldb % 8 != 0 || ldc % 8 != 0) return false;
  if (ep.out_f32 || (ep.dact_pre && (ep.act & EMO_ACT_SAVE_DACT))) return false;
  if (ep.residual && ep.ldr % 8 != 0) return false;
  if ((long)M * lda * 2 >= (1L << 32) || (long)N * ldb * 2 >= (1L << 32)) return false;
  // (N = 512: the transducer joint's data gradient, dz [65 536 x 1024] . W_out^T -- 143 us on the 64 x 64 kernel)
  // (option big_n256: 1 = N = 512 from K = 512 on, N = 256 from K = 2048 on; 2 = N = 256 from K = 512 on as well -- the second
  // feed-forward product, K = 1024: 29.5 against 38 us alone, but 72 against 44 us INSIDE the step, where its operand was written by
  // the launch before: three same-box pairs 29.4 against 28.9 ms per step)
  if (g_big_n256 && M >= 8192 && ((N == 512 && K >= 512) || (N == 256 && K >= (g_big_n256 >= 2 ? 512 : 2048)))) return true;
  if (ep.residual || ep.dact_pre) return false;   // (only measured for the shapes above)
  return (long)cdiv(M, 128) * cdiv(N, 256) >= g_big_min_tiles;
}
void emo_gemm_set_big_n256(int v) { g_big_n256 = v < 0 ? 0 : v; }
int emo_gemm_nt_big_ep(int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                       const emoasr_epilogue_t& ep, hipStream_t s) {
  BigArgs a{};
  a.M = M; a.N = N; a.K = K; a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc;
  a.ep = ep;
  const int bm = g_big_bm ? g_big_bm : pick_bm_plain(M, N, K);
  a.tiles_m = cdiv(M, bm); a.tiles_n = cdiv(N, 256);
  return launch_big_bm<0>(a, bm, a.tiles_m * a.tiles_n, s);
}
void emo_gemm_set_big_min_tiles(int v) { g_big_min_tiles = v; }

// C[M,N] (bf16) = relu?(A[M,K] . B[N,K]^T + bias): N % 8 == 0, K % 64 == 0, 16-byte aligned rows.
extern "C" int emoasr_gemm_nt_big(int dtype, int M, int N, int K, const void* A, long lda, const void* B, long ldb,
                                  void* C, long ldc, const float* bias, int relu, void* stream) {
  EMO_CHECK(dtype == EMO_BF16, "gemm_nt_big: bf16 only");
  EMO_CHECK(M > 0 && N > 0 && N % 8 == 0 && K > 0 && K % 64 == 0, "gemm_nt_big: needs N %% 8 == 0, K %% 64 == 0 (M=%d N=%d K=%d)", M, N, K);
  EMO_CHECK(lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0, "gemm_nt_big: leading dimensions must be multiples of 8");
  EMO_CHECK((long)M * lda * 2 < (1L << 32) && (long)N * ldb * 2 < (1L << 32), "gemm_nt_big: operands must be < 4 GiB");
  emoasr_epilogue_t ep{};
  ep.alpha = 1.f; ep.bias = bias; ep.act = relu ? EMO_ACT_RELU : EMO_ACT_NONE; ep.res_scale = 1.f;
  return emo_gemm_nt_big_ep(M, N, K, A, lda, B, ldb, C, ldc, ep, (hipStream_t)stream);
}

static int rnnt_head_launch(int nrows, int V, int J, const void* h, const void* w, const float* bias, void* C, long ldc,
                            const BigRnnt& rn, hipStream_t s) {
  EMO_CHECK(nrows > 0 && V % 8 == 0 && V >= 64 && J % 64 == 0, "rnnt_head: needs V %% 8 == 0, J %% 64 == 0 (rows=%d V=%d J=%d)", nrows, V, J);
  EMO_CHECK((long)nrows * J * 2 < (1L << 32) && (long)V * J * 2 < (1L << 32), "rnnt_head: operands must be < 4 GiB (chunk the rows)");
  BigArgs a{};
  a.M = nrows; a.N = V; a.K = J; a.A = h; a.lda = J; a.B = w; a.ldb = J; a.C = C; a.ldc = ldc;
  a.ep.alpha = 1.f; a.ep.bias = bias; a.ep.res_scale = 1.f;
  a.rn = rn;
  const int bm = g_big_bm ? g_big_bm : pick_bm_plain(nrows, V, J);
  a.tiles_m = cdiv(nrows, bm); a.tiles_n = cdiv(V, 256);
  return launch_big_bm<0>(a, bm, a.tiles_m * a.tiles_n, s);
}

// Forward of the transducer's output layer over lattice cells row0 .. row0 + nrows (h: their [nrows, J] joint activations):
// part [nrows, ceil(V / 64), 2], zb / zy [nrows] (see BigRnnt).  The logits are never stored.
extern "C" int emoasr_rnnt_head_fwd(int dtype, long row0, int nrows, int Tn, int U, int V, int J, int Lmax, const void* h,
                                    const void* w, const float* bias, const int* labels, const int* ylens, int blank,
                                    float* part, long part_rows, float* zb, float* zy, const int* ycol, void* stream) {
  EMO_CHECK(dtype == EMO_BF16, "rnnt_head_fwd: bf16 only");
  if (nrows == 0) return 0;
  EMO_CHECK(blank >= 0 && blank < V && row0 + nrows < (1L << 31), "rnnt_head_fwd: bad blank / cell range");
  EMO_CHECK(row0 >= 0 && row0 + nrows <= part_rows, "rnnt_head_fwd: cells %ld..%ld outside the partial table's %ld rows", row0, row0 + nrows, part_rows);
  BigRnnt rn{};
  rn.mode = 1; rn.Tn = Tn; rn.U = U; rn.Lmax = Lmax; rn.blank = blank; rn.nchunk = cdiv(V, 64); rn.row0 = row0;
  EMO_CHECK(ycol, "rnnt_head_fwd: ycol (emoasr_rnnt_ycol) required");
  rn.labels = labels; rn.ylens = ylens; rn.part = part; rn.part_rows = part_rows; rn.part_row0 = row0; rn.zb = zb; rn.zy = zy;
  rn.ycol = ycol;
  return rnnt_head_launch(nrows, V, J, h, w, bias, nullptr, V, rn, (hipStream_t)stream);
}

// part [nchunk][rows][2] (chunk-major): one thread per row, 4 chunks per round trip
__global__ __launch_bounds__(64) void lse_parts_kernel(long rows, int nchunk, const float* __restrict__ part, float* __restrict__ lse) {
  const long row = (long)blockIdx.x * 64 + threadIdx.x;
  if (row >= rows) return;
  const float2* pp = reinterpret_cast<const float2*>(part) + row;
  float m = -INFINITY, s = 0.f;
  int c = 0;
  for (; c + 4 <= nchunk; c += 4) {
    float2 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = pp[(long)(c + k) * rows];
    const float mn = fmaxf(fmaxf(fmaxf(v[0].x, v[1].x), fmaxf(v[2].x, v[3].x)), m);
    s = s * __expf(m - mn);
#pragma unroll
    for (int k = 0; k < 4; ++k) s += v[k].y * __expf(v[k].x - mn);
    m = mn;
  }
  for (; c < nchunk; ++c) {
    const float2 v = pp[(long)c * rows];
    const float mn = fmaxf(m, v.x);
    s = s * __expf(m - mn) + v.y * __expf(v.x - mn);
    m = mn;
  }
  lse[row] = m + logf(s);
}

// C[M,N] = A . B^T + bias (bf16) AND lse[m] = log sum_n exp(C[m,n]) of the stored row, in one pass: the soft-max partials leave the
// product's epilogue (part: scratch [ceil(N / 64), M, 2] f32).  The CTC head (decoders/ctc.py:103-113: Linear + log_softmax)
// without the separate pass that read the 703 MB of logits back.
extern "C" int emoasr_gemm_nt_lse(int dtype, int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                                  const float* bias, float* part, float* lse, void* stream) {
  EMO_CHECK(dtype == EMO_BF16, "gemm_nt_lse: bf16 only");
  if (M == 0) return 0;
  EMO_CHECK(N % 8 == 0 && N >= 64 && K % 64 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0, "gemm_nt_lse: needs N %% 8 == 0, K %% 64 == 0");
  EMO_CHECK((long)M * lda * 2 < (1L << 32) && (long)N * ldb * 2 < (1L << 32), "gemm_nt_lse: operands must be < 4 GiB");
  BigArgs a{};
  a.M = M; a.N = N; a.K = K; a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc;
  a.ep.alpha = 1.f; a.ep.bias = bias; a.ep.res_scale = 1.f;
  a.rn.mode = 3; a.rn.nchunk = cdiv(N, 64); a.rn.part = part; a.rn.part_rows = M; a.rn.part_row0 = 0;
  const int bm = g_big_bm ? g_big_bm : pick_bm_plain(M, N, K);
  a.tiles_m = cdiv(M, bm); a.tiles_n = cdiv(N, 256);
  if (launch_big_bm<0>(a, bm, a.tiles_m * a.tiles_n, (hipStream_t)stream)) return 1;
  lse_parts_kernel<<<cdiv(M, 64), 64, 0, (hipStream_t)stream>>>(M, a.rn.nchunk, part, lse);
  EMO_LAUNCH_CHECK();
  return 0;
}

// Gradient rows of the same layer, recomputed: dz[n, :] (bf16, row stride lddz) for the nrows cells whose constants are in
// coef [nrows, 4] (lse, occ, gamma_blank, gamma_label -- already scaled, emoasr_rnnt_coef) and label columns in ycol [nrows].
extern "C" int emoasr_rnnt_head_grad(int dtype, int nrows, int V, int J, const void* h, const void* w, const float* bias,
                                     const float* coef, const int* ycol, int blank, void* dz, long lddz, void* stream) {
  EMO_CHECK(dtype == EMO_BF16, "rnnt_head_grad: bf16 only");
  if (nrows == 0) return 0;
  EMO_CHECK(lddz % 8 == 0 && lddz >= V, "rnnt_head_grad: bad row stride");
  BigRnnt rn{};
  rn.mode = 2; rn.blank = blank; rn.nchunk = cdiv(V, 64); rn.coef = coef; rn.ycol = ycol;
  return rnnt_head_launch(nrows, V, J, h, w, bias, dz, lddz, rn, (hipStream_t)stream);
}

// Conv2d forward through the large-tile kernel; called by emoasr_conv2_fwd (gemm.hip) for bf16, C % 256 == 0.
int emo_conv2_fwd_big(int B, int T1, int F1, int C, const void* y1, const void* w, void* y2, const float* bias,
                      int relu, hipStream_t s) {
  const int T2 = (T1 - 3) / 2 + 1, F2 = (F1 - 3) / 2 + 1;
  EMO_CHECK((long)B * T1 * F1 * C * 2 < (1L << 32), "conv2: input must be < 4 GiB");
  BigArgs a{};
  a.M = B * T2 * F2; a.N = C; a.K = 9 * C; a.A = y1; a.B = w; a.ldb = 9 * C; a.C = y2; a.ldc = C;
  a.bias = bias; a.relu = relu;
  a.cg = BigConv{T1, F1, T2, F2, C};
  a.korder = g_big_korder;
  const int bm = g_big_bm ? g_big_bm : pick_bm(a.M, n_cu_cached());
  a.tiles_m = cdiv(a.M, bm); a.tiles_n = C / 256;
  return launch_big_bm<1>(a, bm, a.tiles_m * a.tiles_n, s);
}

// dy1[b,t1,f1,c] = relu'(y1[b,t1,f1,c]) * sum_{kh,kw,n} dy2[b,(t1-kh)/2,(f1-kw)/2,n] * W[n,c,kh,kw]: the four output-parity
// classes (DgradGeom in gemm.hip) as ONE launch, heaviest class first.  wt: the weight as [c][kh][kw][n] (k-contiguous for
// every tap), i.e. conv.2.weight.permute(1, 2, 3, 0).
extern "C" int emoasr_conv2_dgrad_kc(int dtype, int B, int T1, int F1, int C, const void* dy2, const void* wt,
                                     const void* y1, void* dy1, void* stream) {
  EMO_CHECK(dtype == EMO_BF16, "conv2_dgrad_kc: bf16 only");
  EMO_CHECK(T1 >= 3 && F1 >= 3, "conv2_dgrad_kc: input too small (T1=%d F1=%d)", T1, F1);
  EMO_CHECK(C % 256 == 0, "conv2_dgrad_kc: C must be a multiple of 256");
  const int T2 = (T1 - 3) / 2 + 1, F2 = (F1 - 3) / 2 + 1;
  EMO_CHECK((long)B * T1 * F1 * C * 2 < (1L << 32), "conv2_dgrad_kc: tensors must be < 4 GiB");
  BigArgs a{};
  a.korder = g_big_korder;
  a.N = C; a.K = 4 * C; a.A = dy2; a.B = wt; a.ldb = 9 * C; a.C = dy1; a.ldc = C; a.dmask = y1;
  BigDgrad& g = a.dg;
  g.T1 = T1; g.F1 = F1; g.T2 = T2; g.F2 = F2; g.C = C;
  long rows = 0;
  for (int pt = 0; pt < 2; ++pt)
    for (int pf = 0; pf < 2; ++pf) {
      const int nI = (T1 - pt + 1) / 2, nJ = (F1 - pf + 1) / 2;
      if (nI <= 0 || nJ <= 0) continue;
      const int c = g.ncls++;
      g.pt[c] = pt; g.pf[c] = pf; g.nI[c] = nI; g.nJ[c] = nJ; g.M[c] = B * nI * nJ;
      g.ntap[c] = 0;
      for (int kh = pt; kh < 3; kh += 2)
        for (int kw = pf; kw < 3; kw += 2) {
          const int t = g.ntap[c]++;
          g.dh[c][t] = kh / 2; g.dw[c][t] = kw / 2; g.wtap[c][t] = (kh * 3 + kw) * C;
        }
      rows += (long)g.M[c] * g.ntap[c];
    }
  // tile height from the work-weighted row count (a 4-tap tile runs 4x as long as a 1-tap tile)
  const int bm = g_big_bm ? g_big_bm : pick_bm(rows / 4, n_cu_cached());
  int t0 = 0;
  for (int c = 0; c < g.ncls; ++c) { g.tile0[c] = t0; t0 += cdiv(g.M[c], bm); }
  g.tile0[g.ncls] = t0;
  a.tiles_m = t0; a.tiles_n = C / 256;
  return launch_big_bm<2>(a, bm, t0 * a.tiles_n, (hipStream_t)stream);
}
