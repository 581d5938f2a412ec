// Fused multi-head attention with the Conformer relative-position term, forward and
// backward, for head size DK = 64.  Scores never reach HBM.
//
// Reference semantics (all f32 there):
//   asr/modeling/transformer.py:62-99   MultiHeadedAttention (mask -> finfo.min, softmax,
//                                        masked_fill 0, dropout, .V)
//   asr/modeling/conformer.py:68-95     RelMultiHeadedAttention: ac = (q+u).k^T,
//                                        bd = rel_shift((q+v).p^T)  =>  bd[i,j] = (q_i+v).p[i-j]
//   scores = (ac + bd) / sqrt(dk)
// The rel_shift pad/view/slice dance of the reference is pure index arithmetic here:
// the projected position table `pos` has row r <-> relative offset rel = Tq-1-r, so
// bd[i,j] = (q_i+v) . pos[Tq-1-(i-j)].  For a 32x32 (query, key) tile only a band of 63
// consecutive table rows is needed; the band product is computed with MFMA into a
// 32x64 tile and "skewed" into place through a wave-private LDS buffer.
//
// Design: every wave works alone on a 32-row tile (no block barriers); a block is just
// four independent waves.  k-contiguous MFMA operands are loaded as fragments straight
// from global memory (the per-(b,h) slices are L2 resident); operands whose reduction
// index is the row index in memory (V in forward; K, the band, Q, dO in backward) are
// staged through wave-private LDS and read back transposed (ds_read_b64_tr_b16 in bf16,
// plain b32 reads in f32).  Accumulator tiles are fed to the next MFMA in registers
// ("column on the lane, rows in registers" chaining).
//
//   attn_fwd_kernel   : per 32 queries, loops over key tiles (online softmax)
//   attn_bwd_dq_kernel: per 32 queries, loops over key tiles   -> dq, dbias_u, dbias_v
//   attn_bwd_dkv_kernel: per 32 keys, loops over query tiles   -> dk, dv
//   attn_bwd_dpos_kernel: per 32 table rows, loops over query tiles -> dpos
#include <algorithm>
#include "mma.h"
#include "../../include/emoasr_hip.h"


namespace {

constexpr int DK = 64;

template <typename T> constexpr bool kSplit = std::is_same<T, f32s>::value;   // f32 storage, split-bf16 products (mma.h)
template <typename T> struct AttnCfg {
  static constexpr int NK = DK / Mma<T>::KSTEP;         // k-steps over the head dim (4 / 32)
  static constexpr int NS = 32 / Mma<T>::KSTEP;         // k-steps over a 32-row tile (2 / 16)
  static constexpr int LD = sizeof(T) == 2 ? 72 : (kSplit<T> ? 68 : 64);   // row stride of staged 64-wide LDS tiles (bf16: 144 B
                                                        // rows, split f32: 272 B rows -> conflict-free 16-byte k-contiguous reads)
};

// ---- fragments straight from global memory (k-contiguous operand) -----------------
template <typename T>
__device__ __forceinline__ typename Mma<T>::Frag frag_global(const T* base, long ld, int row, bool valid,
                                                             int kk, int lane, const float* bias) {
  // bounds-checked buffer loads (common.h): an invalid row is an out-of-range offset, not a branch,
  // so the NK fragment loads of a tile are all in flight together.  `base` is per (batch, head):
  // wave-uniform, and one utterance's rows are far below the 4 GiB offset range.
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(base);
  if constexpr (sizeof(T) == 2) {
    const int d0 = kk * 16 + 8 * (lane >> 5);
    bf16x8 f = buf_load16<T>(rs, valid ? (unsigned)(((long)row * ld + d0) * 2) : EMO_OOB).v;
    if (bias) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = (bf16)((float)f[j] + bias[d0 + j]);
    }
    return f;
  } else if constexpr (kSplit<T>) {
    const int d0 = kk * 16 + 8 * (lane >> 5);
    const unsigned off = valid ? (unsigned)(((long)row * ld + d0) * 4) : EMO_OOB;
    const Vec16<T> a = buf_load16<T>(rs, off), b = buf_load16<T>(rs, valid ? off + 16u : EMO_OOB);
    float x[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { x[j] = a.v[j]; x[4 + j] = b.v[j]; }
    if (bias) {
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] += bias[d0 + j];
    }
    return split_regs8(x);
  } else {
    const int d = kk * 2 + (lane >> 5);
    float v = buf_load_f32<T>(rs, valid ? (unsigned)(((long)row * ld + d) * 4) : EMO_OOB);
    if (bias) v += bias[d];
    return v;
  }
}

// the same fragment at a BYTE offset that already holds the row and the lane's half (frag_half_bytes): k-step kk is a constant
// added to it -- no row arithmetic, no guard (the descriptor's size is the guard: make_rsrc_n)
template <typename T> __device__ __forceinline__ unsigned frag_half_bytes(int lane) {
  return (sizeof(T) == 2 ? 16u : (kSplit<T> ? 32u : 4u)) * (unsigned)(lane >> 5);
}
template <typename T>
__device__ __forceinline__ typename Mma<T>::Frag frag_at(__amdgpu_buffer_rsrc_t rs, unsigned off, int kk) {
  if constexpr (sizeof(T) == 2) {
    return buf_load16<T>(rs, off + 32u * kk).v;
  } else if constexpr (kSplit<T>) {
    const Vec16<T> a = buf_load16<T>(rs, off + 64u * kk), b = buf_load16<T>(rs, off + 64u * kk + 16u);
    const float x[8] = {a.v[0], a.v[1], a.v[2], a.v[3], b.v[0], b.v[1], b.v[2], b.v[3]};
    return split_regs8(x);
  } else {
    return buf_load_f32<T>(rs, off + 8u * kk);
  }
}

// ---- stage NROWS x 64 rows into wave-private LDS (row-major, stride LD) --------------
template <typename T, int NROWS>
__device__ __forceinline__ void stage_rows(T* dst, const T* base, long ld, int row0, int row_lo, int row_hi,
                                           int lane, const float* bias) {
  constexpr int VEC = 16 / sizeof(T), PER_ROW = DK / VEC, LD = AttnCfg<T>::LD;
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(base);
#pragma unroll
  for (int v = lane; v < NROWS * PER_ROW; v += 64) {
    const int r = v / PER_ROW, piece = (v % PER_ROW) * VEC;
    const int row = row0 + r;
    const bool ok = row >= row_lo && row < row_hi;
    Vec16<T> x = buf_load16<T>(rs, ok ? (unsigned)(((long)row * ld + piece) * sizeof(T)) : EMO_OOB);
    if (bias) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) x.set(j, x.get(j) + bias[piece + j]);
    }
    store16(dst + r * LD + piece, x);
  }
}

// ---- A operand for "Y = A . X" where X is a 32x32 accumulator tile fed as B ----------
// lds tile is [k = 32 rows][64 cols] (stride LD); returns A[row = col0 + (lane&31)][k-step ks]
// in the permuted k order the accumulator-as-operand trick needs.
template <typename T, bool TR>
__device__ __forceinline__ typename Mma<T>::Frag chain_a(const T* lds, int ks, int col0, int lane) {
  constexpr int LD = AttnCfg<T>::LD;
  if constexpr (sizeof(T) == 2) {
    const int h = lane >> 5;
    if constexpr (TR) {
      union { bf16x8 f; s16x4 q[2]; } u;
      u.q[0] = tr_read4(lds, LD, 16 * ks + 4 * h, col0, lane);
      u.q[1] = tr_read4(lds, LD, 16 * ks + 8 + 4 * h, col0, lane);
      return u.f;
    } else {
      bf16x8 f;
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = lds[(16 * ks + 8 * (j >> 2) + 4 * h + (j & 3)) * LD + col0 + (lane & 31)];
      return f;
    }
  } else if constexpr (kSplit<T>) {
    // the bf16 operand's (permuted) k order, element by element out of the packed tile
    const int h = lane >> 5;
    const unsigned* p = reinterpret_cast<const unsigned*>(lds) + col0 + (lane & 31);
    unsigned w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = p[(16 * ks + 8 * (j >> 2) + 4 * h + (j & 3)) * LD];
    return split_unpack8(w);
  } else {
    return lds[c_row(ks, lane) * LD + col0 + (lane & 31)];
  }
}
template <typename T>
__device__ __forceinline__ typename Mma<T>::Frag chain_b(const f32x16& x, int ks) {
  if constexpr (sizeof(T) == 2) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (bf16)x[8 * ks + j];
    return f;
  } else if constexpr (kSplit<T>) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = x[8 * ks + j];
    return split_regs8(v);
  } else {
    return x[ks];
  }
}

__device__ __forceinline__ void zero16(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

struct HeadPtrs {  // per (b, h) base pointers (element type erased)
  const void *q, *k, *v, *pos, *dout;
  void *out, *dq, *dk, *dv;
  const float *bias_u, *bias_v;
  float *lse, *delta;
  int klen;
};

template <typename T>
__device__ __forceinline__ HeadPtrs head_ptrs(const emoasr_attn_t& a, int b, int h) {
  HeadPtrs p;
  const long ho = (long)h * DK;
  p.q = (const T*)a.q + (long)b * a.Tq * a.ldq + ho;
  p.k = (const T*)a.k + (long)b * a.Tk * a.ldk + ho;
  p.v = (const T*)a.v + (long)b * a.Tk * a.ldv + ho;
  p.pos = a.pos ? (const T*)a.pos + ho : nullptr;
  p.dout = a.dout ? (const T*)a.dout + (long)b * a.Tq * a.ldo + ho : nullptr;
  p.out = a.out ? (T*)a.out + (long)b * a.Tq * a.ldo + ho : nullptr;
  p.dq = a.dq ? (T*)a.dq + (long)b * a.Tq * a.ldq + ho : nullptr;
  p.dk = a.dk ? (T*)a.dk + (long)b * a.Tk * a.ldk + ho : nullptr;
  p.dv = a.dv ? (T*)a.dv + (long)b * a.Tk * a.ldv + ho : nullptr;
  p.bias_u = a.bias_u ? a.bias_u + ho : nullptr;
  p.bias_v = a.bias_v ? a.bias_v + ho : nullptr;
  p.lse = a.lse + ((long)b * a.H + h) * a.Tq;
  p.delta = a.delta ? a.delta + ((long)b * a.H + h) * a.Tq : nullptr;
  p.klen = a.klens ? min(a.klens[b], a.Tk) : a.Tk;
  return p;
}

// Stacked micro-batches (emoasr_attn_t::nseg > 1): a workgroup first turns the launch's arguments into those of ITS segment --
// base pointers moved to the segment's first row / table row, B / Tq / Tk its own, the utterance index made local, the dropout
// seed offset -- and everything below runs as for a dense batch.  All of it is wave-uniform (scalar registers).
// (The tables are indexed dynamically only in the KERNEL ARGUMENT -- constant memory; the private copy that seg_apply edits
// is touched through fixed fields only, so it stays in scalar registers instead of scratch.)
struct SegRef { int s, b0, nb, T; long row, prow; };
__device__ __forceinline__ SegRef seg_ref(const emoasr_attn_t& karg, int s) {
  return SegRef{s, karg.seg_b0[s], karg.seg_b0[s + 1] - karg.seg_b0[s], karg.seg_T[s], karg.seg_row[s], karg.seg_prow[s]};
}
__device__ __forceinline__ SegRef seg_of_utt(const emoasr_attn_t& karg, int b) {
  int s = 0;
  for (int k = 1; k < EMOASR_MAX_SEGMENTS; ++k) s += (k < karg.nseg && b >= karg.seg_b0[k]) ? 1 : 0;
  return seg_ref(karg, s);
}
// workgroup slot z of a stacked launch -> (segment, utterance inside it): the slots run through the segments in seg_order
// (longest first, filled by the launcher), so the hardware's in-order dispatch is a longest-processing-time-first schedule
__device__ __forceinline__ SegRef seg_of_slot(const emoasr_attn_t& karg, int z, int& local) {
  int v = z, s = karg.seg_order[0];
  bool found = false;
  for (int k = 0; k < EMOASR_MAX_SEGMENTS; ++k) {
    if (k < karg.nseg && !found) {
      const int sk = karg.seg_order[k], nbk = karg.seg_b0[sk + 1] - karg.seg_b0[sk];
      if (v < nbk) { s = sk; found = true; }
      else v -= nbk;
    }
  }
  local = v;
  return seg_ref(karg, s);
}
// Workgroup coordinates of the forward / fused-backward launches.  nt == 0: the plain 3-D grid (tile, head, utterance slot).
// nt > 0 (a 1-D launch of 8 * ceil(H B / 8) * nt workgroups): the nt tiles of one (head, utterance) GROUP -- which read the same
// K / V (forward) or Q / dO / position rows (backward) -- get hardware ids of one residue class mod 8, i.e. one XCD and one L2,
// next to each other in dispatch order; the groups go round-robin over the XCDs in slot order (longest segment first).
struct Blk3 { int x, y, z; bool ok; };
__device__ __forceinline__ Blk3 attn_block(int nt, int H, int B) {   // (H, B: the grid's y and z extents)
  if (nt <= 0) return Blk3{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, true};
  const int lin = blockIdx.x, xcd = lin & 7, slot = lin >> 3;
  const int gl = slot / nt, tile = slot - gl * nt, g = gl * 8 + xcd;
  return Blk3{tile, g % H, g / H, g < H * B};
}
__device__ __forceinline__ SegRef seg_of_row(const emoasr_attn_t& karg, long row) {
  int s = 0;
  for (int k = 1; k < EMOASR_MAX_SEGMENTS; ++k) s += (k < karg.nseg && row >= karg.seg_row[k]) ? 1 : 0;
  return seg_ref(karg, s);
}
template <typename T>
__device__ __forceinline__ void seg_apply(emoasr_attn_t& a, const SegRef& g) {
  const long r = g.row;
  a.q = (const T*)a.q + r * a.ldq; a.k = (const T*)a.k + r * a.ldk; a.v = (const T*)a.v + r * a.ldv;
  if (a.out) a.out = (T*)a.out + r * a.ldo;
  if (a.dout) a.dout = (const T*)a.dout + r * a.ldo;
  if (a.dq) a.dq = (T*)a.dq + r * a.ldq;
  if (a.dk) a.dk = (T*)a.dk + r * a.ldk;
  if (a.dv) a.dv = (T*)a.dv + r * a.ldv;
  if (a.pos) a.pos = (const T*)a.pos + g.prow * a.ldp;
  if (a.dpos) a.dpos += g.prow * (long)(a.H * DK);
  a.lse += r * a.H;
  if (a.delta) a.delta += r * a.H;
  if (a.klens) a.klens += g.b0;
  if (a.keep_mask) a.keep_mask += r * a.H * a.keep_nw;
  a.B = g.nb;
  a.Tq = a.Tk = g.T;
  a.seed += 0x9E3779B97F4A7C15ull * (uint64_t)g.s;
}

// ------------------------------------------------------------------------------------
// Score tile.  `row_*` fragments index the tile's rows, `col_*` its columns:
//   SWAPPED  (forward, dq):   rows = keys j0.., cols = queries i0..   (row side = K / band)
//   !SWAPPED (dkv, dpos):     rows = queries i0.., cols = keys j0..   (row side = Q)
// Returns raw ac + bd (unscaled) in `acc` (C layout).
// Band rows: r = rbase + c, c in [0,64), rbase = Tq - 32 - i0 + j0; element (i_l, j_l) uses
// c = 31 - i_l + j_l.
// ------------------------------------------------------------------------------------
template <typename T, bool SWAPPED>
__device__ __forceinline__ void score_tile(f32x16& acc, const emoasr_attn_t& a, const HeadPtrs& hp, int i0,
                                           int j0, const typename Mma<T>::Frag* qu,
                                           const typename Mma<T>::Frag* qv,
                                           const typename Mma<T>::Frag* kf, float* Gs, int lane) {
  using M_ = Mma<T>;
  constexpr int NK = AttnCfg<T>::NK;
  const T* kbase = (const T*)hp.k;
  zero16(acc);
  if constexpr (SWAPPED) {
    const int krow = j0 + (lane & 31);
    const bool kval = krow >= 0 && krow < a.Tk;
#pragma unroll
    for (int kk = 0; kk < NK; ++kk)
      acc = M_::mma(frag_global<T>(kbase, a.ldk, krow, kval, kk, lane, nullptr), qu[kk], acc);
  } else {
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) acc = M_::mma(qu[kk], kf[kk], acc);
  }
  if (hp.pos == nullptr) return;
  const T* pbase = (const T*)hp.pos;
  const int rbase = a.Tq - 32 - i0 + j0, rmax = 2 * a.Tq - 2;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    f32x16 g;
    zero16(g);
    const int prow = clampi(rbase + 32 * ct + (lane & 31), 0, rmax);
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      const typename M_::Frag pf = frag_global<T>(pbase, a.ldp, prow, true, kk, lane, nullptr);
      if constexpr (SWAPPED) g = M_::mma(pf, qv[kk], g);   // g[c][i]
      else g = M_::mma(qv[kk], pf, g);                     // g[i][c]
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if constexpr (SWAPPED) Gs[(32 * ct + c_row(r, lane)) * 32 + (lane & 31)] = g[r];
      else Gs[c_row(r, lane) * 64 + 32 * ct + (lane & 31)] = g[r];
    }
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if constexpr (SWAPPED) {  // row = key j_l, col = query i_l
      const int jl = c_row(r, lane), il = lane & 31;
      acc[r] += Gs[(31 - il + jl) * 32 + il];
    } else {                  // row = query i_l, col = key j_l
      const int il = c_row(r, lane), jl = lane & 31;
      acc[r] += Gs[il * 64 + 31 - il + jl];
    }
  }
  __builtin_amdgcn_wave_barrier();
}

#ifdef EMO_ATTN_STAMP
__device__ unsigned long long g_fwd_stamps[64 * 8];
#define EMO_FSTAMP(k)                                                                                                   \
  do {                                                                                                                  \
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 && j0 / 32 < 64)                       \
      g_fwd_stamps[(j0 / 32) * 8 + (k)] = __builtin_amdgcn_s_memtime();                                                 \
  } while (0)
#else
#define EMO_FSTAMP(k) do {} while (0)
#endif

// element index of the attention-dropout mask: rows of an EVEN stride, so that keys 2m and 2m + 1 of a row are one hash pair
// (common.h: dropout_keep2)
__device__ __forceinline__ uint64_t drop_index(const emoasr_attn_t& a, int b, int h, int i, int j) {
  return (((uint64_t)b * a.H + h) * a.Tq + i) * (uint64_t)((a.Tk + 1) & ~1) + j;
}

// Sum over the 32 lanes of each half wave without the LDS pipeline: four DPP adds inside the rows of 16 (quad swaps, half mirror, mirror),
// then row_bcast:15 carries row 0's total into row 1 (row 2's into row 3).  The total is valid in lanes 16..31 of each half -- callers read
// lane il == 31.  (__shfl_xor is a ds_bpermute: the five-step trees over 32 accumulator registers were 160 LDS round trips per wave and the
// bias-gradient epilogues took 20 k cycles of a 72 k cycle workgroup.)
template <int CTRL, int ROWS = 0xf> __device__ __forceinline__ float dpp_take(const float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWS, 0xf, false));
}
// the other half wave's value in every lane, without the LDS pipeline: v_permlane32_swap with both operands = v leaves (lower half's
// value, upper half's value) in the two results of every lane (__shfl_xor(v, 32) is a ds_bpermute round trip -- two per key tile on the
// critical path of the forward's online soft-max)
// bit `K` of a mask word as 0 / -1 in ONE instruction (v_bfe_i32 with width 1 sign-extends).  Written as inline assembly: with a
// compile-time bit index the compiler rewrites `sbfe(w, K, 1) & keep` into and + compare + select (three instructions per element).
template <int K> __device__ __forceinline__ unsigned bit_mask(const unsigned w) {
  unsigned r;
  asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(r) : "v"(w), "n"(K));
  return r;
}
// the keep bit of accumulator row r (key 8 (r >> 2) + (r & 3) of this half wave's byte-shifted word) as 0 / -1
__device__ __forceinline__ unsigned bit_of_row(const unsigned w, const int r) {
  switch (r) {
#define EMO_BR(R_) case R_: return bit_mask<8 * (R_ >> 2) + (R_ & 3)>(w);
    EMO_BR(0) EMO_BR(1) EMO_BR(2) EMO_BR(3) EMO_BR(4) EMO_BR(5) EMO_BR(6) EMO_BR(7)
    EMO_BR(8) EMO_BR(9) EMO_BR(10) EMO_BR(11) EMO_BR(12) EMO_BR(13) EMO_BR(14) EMO_BR(15)
#undef EMO_BR
  }
  return 0u;
}
__device__ __forceinline__ float xhalf_max(const float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(const float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// sum over groups of 8 lanes on DPP: quad swaps + row_half_mirror
__device__ __forceinline__ float sum8(float v) {
  v += dpp_take<0xB1>(v);
  v += dpp_take<0x4E>(v);
  v += dpp_take<0x141>(v);
  return v;
}
__device__ __forceinline__ float half_sum32(float v) {
  v += dpp_take<0xB1>(v);          // quad_perm [1, 0, 3, 2]
  v += dpp_take<0x4E>(v);          // quad_perm [2, 3, 0, 1]
  v += dpp_take<0x141>(v);         // row_half_mirror
  v += dpp_take<0x140>(v);         // row_mirror: every lane of a row holds the row's sum
  v += dpp_take<0x142, 0xA>(v);    // row_bcast:15 into rows 1 and 3
  return v;
}


// write a [64 d][32 cols] transposed accumulator pair (rows d in registers, col = row index
// of the destination matrix on the lane) to dst[row = r0 + (lane&31)][d], optionally summed
// with a second pair.
template <typename T>
__device__ __forceinline__ void store_dT(T* dst, long ld, int r0, int rlimit, const f32x16* x,
                                         float mul, int lane) {
  const int row = r0 + (lane & 31);
  if (row >= rlimit) return;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = x[dt][4 * g + e] * mul;
      T* p = dst + (long)row * ld + 32 * dt + 8 * g + 4 * (lane >> 5);
      if constexpr (sizeof(T) == 2) {
        bf16x4 v;
        v[0] = (bf16)o[0]; v[1] = (bf16)o[1]; v[2] = (bf16)o[2]; v[3] = (bf16)o[3];
        *reinterpret_cast<bf16x4*>(p) = v;
      } else {
        *reinterpret_cast<f32x4*>(p) = f32x4{o[0], o[1], o[2], o[3]};
      }
    }
}

// ====================================================================================
// forward
// ====================================================================================
// PF2: two prefetch register sets (a full tile of distance, ~390 registers: one wave per SIMD) for grids of
// at most one block per CU; otherwise one set, refilled as soon as the score MFMAs have read it, and two
// blocks per CU (bf16).  Measured at B 20, T' 340 (240 blocks): 36 us with two sets, 45 us with one; at 264
// blocks the two-set kernel's second round of blocks costs 80 us against 62.
// bf16 / f32: the V tile takes the skew tile's bytes (staged after the skew has been read) and the lower band half is reused from
// the previous tile's registers; the split type keeps V in a region of its own, staged at the top of the tile, and loads both band
// halves -- holding V's and the band's (hi, lo) registers across the score phase put it into scratch (288-320 registers + 0.5-1 KB
// of private memory: the f32x3 forward ran 3.4 x slower; the exact f32 one gained 15 % from the new order)
template <typename T> constexpr int fwd_wave_bytes() {
  constexpr int g = 32 * 68 * 4, v = 32 * AttnCfg<T>::LD * (int)sizeof(T);
  return !kSplit<T> ? (g > v ? g : v) : g + v;
}
template <typename T, bool TR, bool PF2>
__global__ __launch_bounds__(256, (sizeof(T) == 2 && !PF2) ? 2 : 1) void attn_fwd_kernel(const emoasr_attn_t a_in, const int nt, const int ks) {
  using M_ = Mma<T>;
  constexpr int NK = AttnCfg<T>::NK, NS = AttnCfg<T>::NS, LD = AttnCfg<T>::LD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  emoasr_attn_t a = a_in;
  const Blk3 blk = attn_block(nt, a_in.H, a_in.B);
  if (!blk.ok) return;
  int b = blk.z;
  if (a_in.nseg > 1) {
    const SegRef g = seg_of_slot(a_in, blk.z, b);
    seg_apply<T>(a, g);
  }
  // ks == 1: 1, 2 or 4 independent waves per workgroup, a query tile each.  ks == 4 (small launches: a batch-1 decode has 40
  // query tiles for 256 CUs): the four waves share ONE query tile and take every fourth key tile; their soft-max states meet in LDS
  const int i0 = (ks > 1 ? blk.x : blk.x * (blockDim.x >> 6) + wave) * 32, h = blk.y;
  if (i0 >= a.Tq) return;
  // wave-private LDS: the f32 skew tile G[query][band column] ([32][LDG]) and, AFTER the skew has been read, the V tile in the same
  // bytes (round 6: 8.7 KB per wave instead of 12.6 -- the stacked launches' one-wave workgroups were LDS-limited to 12 per CU)
  constexpr int LDG = 68;
  constexpr int WAVE_BYTES = fwd_wave_bytes<T>();
  constexpr bool LEAN = !kSplit<T>;
  float* Gs = reinterpret_cast<float*>(smem + wave * WAVE_BYTES);
  T* Vs = reinterpret_cast<T*>(smem + wave * WAVE_BYTES + (LEAN ? 0 : 32 * LDG * 4));

  const HeadPtrs hp = head_ptrs<T>(a, b, h);
  const int qi = i0 + (lane & 31);
  const bool qval = qi < a.Tq;
  typename M_::Frag qu[NK], qv[NK];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    qu[kk] = frag_global<T>((const T*)hp.q, a.ldq, qi, qval, kk, lane, hp.bias_u);
    qv[kk] = hp.pos ? frag_global<T>((const T*)hp.q, a.ldq, qi, qval, kk, lane, hp.bias_v) : qu[kk];
  }
  float m = -INFINITY, l = 0.f;
  f32x16 o[2];
  zero16(o[0]); zero16(o[1]);
  int kend = hp.klen;
  if (a.causal) kend = min(kend, i0 + 32);

  // Operand prefetch: the K / position-band fragments and the V rows of key tile j0+32 are loaded (bounds-
  // checked buffer loads, no branches) as soon as tile j0's score MFMAs have read theirs, and fly during its
  // skew, softmax and P.V.  With
  // one wave per SIMD at the L2 batch size, loading each tile right where it was needed exposed two global
  // round trips per tile.
  constexpr int VEC = 16 / sizeof(T), PER_ROW = DK / VEC, VR = 32 * PER_ROW / 64;
  struct Pre { typename M_::Frag kf[NK]; typename M_::Frag pf[2][NK]; Vec16<T> vr[VR]; unsigned mw; };
  const bool rel = hp.pos != nullptr;
  // Operand descriptors sized to the valid rows (keys < kend, table rows < 2 Tq - 1): rows past them -- the last tile's tail, the
  // prefetch behind the last tile, band rows of (query, key) pairs outside the utterance -- read zeros by the descriptor, so a
  // fetch is 16 loads at lane-constant offsets + one stride multiple per operand: no compare / select / 64-bit multiply per load
  // (round 5: 34 exec-mask regions around these loads)
  constexpr unsigned ESZ = sizeof(T);
  const unsigned kstride = (unsigned)a.ldk * ESZ, vstride = (unsigned)a.ldv * ESZ, pstride = (unsigned)a.ldp * ESZ;
  const __amdgpu_buffer_rsrc_t rsK = make_rsrc_n(hp.k, kend > 0 ? (unsigned)(kend - 1) * kstride + DK * ESZ : 0u);
  const __amdgpu_buffer_rsrc_t rsV = make_rsrc_n(hp.v, kend > 0 ? (unsigned)(kend - 1) * vstride + DK * ESZ : 0u);
  const __amdgpu_buffer_rsrc_t rsP = make_rsrc_n(rel ? hp.pos : hp.k, rel ? (unsigned)(2 * a.Tq - 2) * pstride + DK * ESZ : 0u);
  const unsigned k_lane = (unsigned)(lane & 31) * kstride + frag_half_bytes<T>(lane);
  const unsigned p_lane = (unsigned)(lane & 31) * pstride + frag_half_bytes<T>(lane);
  unsigned v_lane[VR];
#pragma unroll
  for (int i = 0; i < VR; ++i) {
    const int v = lane + 64 * i;
    v_lane[i] = (unsigned)(v / PER_ROW) * vstride + (unsigned)((v % PER_ROW) * VEC) * ESZ;
  }
  // the attention-dropout keep mask as bits (emoasr_attn_dropmask hashed it once for this forward AND the backward): one word per
  // (query row, head, 32-key tile), fetched with the tile's operands; without it the mask is hashed inline (dropout_keep2)
  const bool mbits = a.keep_mask != nullptr && a.drop_p > 0.f;
  const __amdgpu_buffer_rsrc_t rsM = make_rsrc(a.keep_mask);
  const unsigned mrow = (unsigned)((((long)b * a.Tq + qi) * a.H + h) * a.keep_nw * 4);
  // `prev`: the register set that holds the band of key tile j0 - 32 ks... only for ks == 1 (consecutive tiles): the band of tile
  // j0 is rows rbase(j0) + [0, 64) and rbase moves by 32 per tile, so its lower half IS the previous tile's upper half -- four
  // fragment loads per tile instead of eight
  auto fetch = [&](Pre& p, int j0, const Pre* prev) {
    const unsigned ko = k_lane + (unsigned)j0 * kstride;
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) p.kf[kk] = frag_at<T>(rsK, ko, kk);
    if (rel) {
      // band row of lane l of tile ct: Tq - 32 - i0 + j0 + 32 ct + l (negative: the offset wraps out of range); nothing past kend
      const unsigned po = (unsigned)((a.Tq - 32 - i0 + j0) * (int)pstride) + p_lane + (j0 < kend ? 0u : 0x80000000u);
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        if constexpr (LEAN) p.pf[0][kk] = prev ? prev->pf[1][kk] : frag_at<T>(rsP, po, kk);
        else p.pf[0][kk] = frag_at<T>(rsP, po, kk);
        p.pf[1][kk] = frag_at<T>(rsP, po + 32u * pstride, kk);
      }
    }
#pragma unroll
    for (int i = 0; i < VR; ++i) p.vr[i] = buf_load16<T>(rsV, v_lane[i] + (unsigned)j0 * vstride);
    p.mw = mbits ? __builtin_amdgcn_raw_buffer_load_b32(rsM, (j0 < kend && qval) ? mrow + 4u * (unsigned)(j0 >> 5) : EMO_OOB, 0, 0) : 0u;
  };
  constexpr bool FAST = sizeof(T) == 2;   // bf16: soft-max in the exp2 domain (one multiply less per element, no -inf tests per element)
  const float c2 = a.scale * 1.4426950408889634f;
  const unsigned keep_bits = __float_as_uint(a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f);
  auto tile = [&](Pre& cur, Pre& nxt, int j0) {
    EMO_FSTAMP(0);
    if constexpr (PF2) fetch(nxt, j0 + 32 * ks, ks == 1 ? &cur : nullptr);
    EMO_FSTAMP(1);
    if constexpr (!LEAN) {
#pragma unroll
      for (int i = 0; i < VR; ++i) {
        const int v = lane + 64 * i;
        lds_stage16(Vs + (v / PER_ROW) * LD + (v % PER_ROW) * VEC, cur.vr[i]);
      }
    }
    // S^T = K . (Q+u)^T + skew(pos_band . (Q+v)^T)   (rows keys, cols queries; see score_tile)
    f32x16 s;
    zero16(s);
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) s = M_::mma(cur.kf[kk], qu[kk], s);
    if (rel) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        f32x16 g;
        zero16(g);
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) g = M_::mma(cur.pf[ct][kk], qv[kk], g);  // g[c][i]
        // G[query il][c]: accumulator registers 4 q .. 4 q + 3 are four consecutive band columns -> one 16-byte store
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(Gs + (lane & 31) * LDG + 32 * ct + 8 * q + 4 * (lane >> 5)) = f32x4{g[4 * q], g[4 * q + 1], g[4 * q + 2], g[4 * q + 3]};
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] += Gs[(lane & 31) * (LDG - 1) + 31 + c_row(r, lane)];   // G[il][31 - il + jl]
      __builtin_amdgcn_wave_barrier();
    }
    EMO_FSTAMP(2);
    // the V tile takes the skew tile's place
    if constexpr (LEAN) {
#pragma unroll
      for (int i = 0; i < VR; ++i) {
        const int v = lane + 64 * i;
        lds_stage16(Vs + (v / PER_ROW) * LD + (v % PER_ROW) * VEC, cur.vr[i]);
      }
    }
    const unsigned mw = cur.mw >> (4 * (lane >> 5));   // this half wave's keys are 8 g + 4 hh + e
    // one register set: every prefetched register has been consumed, refill them with tile j0+32
    if constexpr (!PF2) fetch(cur, j0 + 32 * ks, ks == 1 ? &cur : nullptr);   // (ks > 1: this wave's next key tile is ks tiles on)
    EMO_FSTAMP(3);
    if (a.st && qval) {  // keep the scaled scores for the backward pass (32 queries contiguous per key row)
      float* srow = a.st + (((long)b * a.H + h) * a.Tk + j0) * a.ldst + qi;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kl = c_row(r, lane);
        if (j0 + kl < a.Tk) srow[(long)kl * a.ldst] = s[r] * a.scale;
      }
    }
    EMO_FSTAMP(4);
    float mt = -INFINITY;
    const float sc = FAST ? c2 : a.scale;
    if (j0 + 32 <= hp.klen && !a.causal) {  // (wave-uniform) a tile without masked keys: no per-element range tests
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] *= sc;
        mt = fmaxf(mt, s[r]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kj = j0 + c_row(r, lane);
        const bool masked = kj >= hp.klen || (a.causal && kj > qi);
        s[r] = masked ? -INFINITY : s[r] * sc;
        mt = fmaxf(mt, s[r]);
      }
    }
    mt = xhalf_max(mt);
    const float mn = fmaxf(m, mt);
    float alpha, rs = 0.f;
    if constexpr (FAST) {
      // m, mn in the exp2 domain; a row without a valid key so far keeps mn = -inf: subtract 0 instead (exp2(-inf) = 0 either way)
      const float mref = (mn == -INFINITY) ? 0.f : mn;
      alpha = __builtin_amdgcn_exp2f(m - mref);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(s[r] - mref);
        rs += p;
        s[r] = p;
      }
    } else {
      alpha = (m == -INFINITY) ? 0.f : __expf(m - mn);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = (s[r] == -INFINITY) ? 0.f : __expf(s[r] - mn);
        rs += p;
        s[r] = p;
      }
    }
    rs = xhalf_sum(rs);
    l = l * alpha + rs;
    m = mn;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
    if (a.drop_p > 0.f) {
      if (mbits) {
        // bit 8 g + e of the half wave's word: 0 / -1 -> 0 / keep
#pragma unroll
        for (int r = 0; r < 16; ++r)
          s[r] *= __uint_as_float((unsigned)__builtin_amdgcn_sbfe(mw, 8 * (r >> 2) + (r & 3), 1) & keep_bits);
      } else {
        // accumulator rows 4 g .. 4 g + 3 are keys j0 + 8 g + 4 (lane >> 5) + 0 .. 3: two hash pairs (row base and j0 are even)
        const float keep = 1.f / (1.f - a.drop_p);
        const uint32_t thr = dropout_thr(a.drop_p);
        const uint64_t dbase = drop_index(a, b, h, qi, j0 + 4 * (lane >> 5));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint64_t pr = (dbase + (uint64_t)(8 * g)) >> 1;
          bool k0, k1, k2, k3;
          dropout_keep2(a.seed, pr, thr, k0, k1);
          dropout_keep2(a.seed, pr + 1, thr, k2, k3);
          s[4 * g] *= k0 ? keep : 0.f; s[4 * g + 1] *= k1 ? keep : 0.f;
          s[4 * g + 2] *= k2 ? keep : 0.f; s[4 * g + 3] *= k3 ? keep : 0.f;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    EMO_FSTAMP(5);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int ks = 0; ks < NS; ++ks)
        o[dt] = M_::mma(chain_a<T, TR>(Vs, ks, 32 * dt, lane), chain_b<T>(s, ks), o[dt]);
    __builtin_amdgcn_wave_barrier();
    EMO_FSTAMP(6);
  };
  const int jstep = 32 * ks, jfirst = ks > 1 ? 32 * wave : 0;
  if constexpr (PF2) {
    Pre pa, pb;
    fetch(pa, jfirst, nullptr);
    for (int j0 = jfirst; j0 < kend; j0 += 2 * jstep) {
      tile(pa, pb, j0);
      if (j0 + jstep < kend) tile(pb, pa, j0 + jstep);
    }
  } else {
    Pre pc;
    fetch(pc, jfirst, nullptr);
    for (int j0 = jfirst; j0 < kend; j0 += jstep) tile(pc, pc, j0);
  }
  if (ks > 1) {
    // merge the four waves' (m, l, O) of the same 32 queries: every register of a lane belongs to query lane & 31, so the
    // states combine lane by lane.  Each wave parks its O in its own Gs image (2048 floats) and m / l at the head of its Vs.
    float* sml = Gs + 2048;   // (behind the parked O image: 64 floats)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) Gs[(dt * 16 + r) * 64 + lane] = o[dt][r];
    if (lane < 32) { sml[lane] = m; sml[32 + lane] = l; }
    __syncthreads();
    if (wave != 0) return;
    float mw[4], lw[4], sc[4], mm = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float* q = reinterpret_cast<const float*>(smem + w * WAVE_BYTES) + 2048;
      mw[w] = q[lane & 31]; lw[w] = q[32 + (lane & 31)];
      mm = fmaxf(mm, mw[w]);
    }
    l = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      sc[w] = mw[w] == -INFINITY ? 0.f : (FAST ? __builtin_amdgcn_exp2f(mw[w] - mm) : __expf(mw[w] - mm));
      l += lw[w] * sc[w];
    }
    m = mm;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) acc += reinterpret_cast<const float*>(smem + w * WAVE_BYTES)[(dt * 16 + r) * 64 + lane] * sc[w];
        o[dt][r] = acc;
      }
  }
  const float inv = l > 0.f ? 1.f / l : 0.f;
  store_dT<T>((T*)hp.out, a.ldo, i0, a.Tq, o, inv, lane);
  if (lane < 32 && qval) hp.lse[qi] = l > 0.f ? (FAST ? m * 0.6931471805599453f : m) + __logf(l) : -INFINITY;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Forward, block-staged (round 6; bf16, relative positions, no causal mask / stored scores): the training launches.
// attn_fwd_kernel above lets every wave fetch its own K / band / V fragments straight from global memory: 12-16 loads per wave and
// key tile whose lanes sit in 32 different rows (32 cache lines touched for 1 KB of payload) -- in-kernel stamps put 40-45 % of a
// tile's cycles into issuing them and waiting for them (profiles/r06_attn_phases.txt), with 12 KB per wave and tile crossing the
// CU's vector-memory path.  Here the four waves of a workgroup take FOUR CONSECUTIVE query tiles of one (utterance, head) and
// share the key tile: K and V (32 rows each) and the union of their position bands are staged once per step through LDS by
// coalesced 16-byte pieces (8 lanes per 128-byte row), registers one step ahead.  The band union moves up by 32 rows per key
// tile, so it is a ring of FW + 2 blocks of 32 rows and a step fetches ONE new block: 3 loads per thread and step, 12 KB per
// WORKGROUP and step instead of 48.  Per-wave work is the old kernel's: S^T = K (Q+u)^T + skew(band (Q+v)^T) with the query on
// the lane, online soft-max in the exp2 domain, keep-mask bits (or the inline hash), P chained into O^T += V^T P.
// ------------------------------------------------------------------------------------------------------------------------------------
template <int FW> struct Fwd4Cfg {
  static constexpr int LD = AttnCfg<bf16>::LD, LDG = 68, NRING = FW + 2;
  static constexpr int STAGE_ROWS = 64 + 32 * NRING;                   // K, V, band ring
  static constexpr int GS_BYTES = 32 * LDG * 4;                        // per wave: f32 skew tile
  static constexpr int smem() { return STAGE_ROWS * LD * 2 + FW * GS_BYTES; }
};
// DM: the dropout mode as a template parameter (0 none, 1 keep-mask bits, 2 inline hash) -- as run-time tests of `a.drop_p` and
// `a.keep_mask` they were scalar branches inside the key loop, which cut the step into blocks nothing is scheduled across
template <bool TR, int FW, int DM>
__global__ __launch_bounds__(64 * FW, 2) void attn_fwd4_kernel(const emoasr_attn_t a_in, const int nt) {
  using T = bf16;
  using M_ = Mma<T>;
  using C_ = Fwd4Cfg<FW>;
  constexpr int NK = AttnCfg<T>::NK, NS = AttnCfg<T>::NS, LD = C_::LD, LDG = C_::LDG, NRING = C_::NRING;
  constexpr int VEC = 8, PER_ROW = DK / VEC, NTHR = 64 * FW;
  static_assert(NTHR / PER_ROW == 32, "one piece per thread covers one 32-row tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), il = lane & 31, hh = lane >> 5;
  emoasr_attn_t a = a_in;
  const Blk3 blk = attn_block(nt, a_in.H, a_in.B);
  if (!blk.ok) return;
  int b = blk.z;
  if (a_in.nseg > 1) {
    const SegRef g = seg_of_slot(a_in, blk.z, b);
    seg_apply<T>(a, g);
  }
  const int iblk = blk.x * (32 * FW), h = blk.y;
  if (iblk >= a.Tq) return;   // (a shorter segment of a stacked launch: the grid follows the longest)
  const HeadPtrs hp = head_ptrs<T>(a, b, h);
  const int i0 = iblk + 32 * wave;
  const bool live = i0 < a.Tq;   // a dead wave still stages and joins the barriers
  const int qi = i0 + il;
  const bool qval = qi < a.Tq;

  T* Ks = reinterpret_cast<T*>(smem);
  T* Vs = Ks + 32 * LD;
  T* ring = Ks + 64 * LD;   // block n = table rows Tq - 32 FW - iblk + 32 n + [0, 32) in slot n mod NRING
  float* Gs = reinterpret_cast<float*>(smem + C_::STAGE_ROWS * LD * 2 + wave * C_::GS_BYTES);

  const int kend = hp.klen, nstep = (kend + 31) / 32;
  const unsigned kstride = (unsigned)a.ldk * 2u, vstride = (unsigned)a.ldv * 2u, pstride = (unsigned)a.ldp * 2u;
  // descriptors sized to the valid rows (keys < klen, table rows < 2 Tq - 1): see attn_fwd_kernel
  const __amdgpu_buffer_rsrc_t rsK = make_rsrc_n(hp.k, kend > 0 ? (unsigned)(kend - 1) * kstride + DK * 2u : 0u),
                               rsV = make_rsrc_n(hp.v, kend > 0 ? (unsigned)(kend - 1) * vstride + DK * 2u : 0u),
                               rsP = make_rsrc_n(hp.pos, (unsigned)(2 * a.Tq - 2) * pstride + DK * 2u);
  const int trow = tid / PER_ROW, piece = (tid % PER_ROW) * VEC;
  const unsigned k_lane = (unsigned)trow * kstride + (unsigned)piece * 2u, v_lane = (unsigned)trow * vstride + (unsigned)piece * 2u;
  const unsigned p_lane = (unsigned)((a.Tq - 32 * FW - iblk + trow) * (int)pstride) + (unsigned)piece * 2u;   // block 0
  Vec16<T> pre[3];
  auto fetch = [&](const int step) {   // K / V tile of `step` and the band block it adds (block step + FW)
    const unsigned dead = step < nstep ? 0u : 0x80000000u;
    pre[0] = buf_load16<T>(rsK, (k_lane + (unsigned)(32 * step) * kstride) | dead);
    pre[1] = buf_load16<T>(rsV, (v_lane + (unsigned)(32 * step) * vstride) | dead);
    pre[2] = buf_load16<T>(rsP, (p_lane + (unsigned)(32 * (step + FW)) * pstride) | dead);
  };
  auto stash = [&](const int step) {
    store16(Ks + trow * LD + piece, pre[0]);
    store16(Vs + trow * LD + piece, pre[1]);
    store16(ring + (((step + FW) % NRING) * 32 + trow) * LD + piece, pre[2]);
  };
  // stationary operands of this wave's query tile (B operands: query on the lane), biases added on the way in
  typename M_::Frag qu[NK], qv[NK];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    qu[kk] = frag_global<T>((const T*)hp.q, a.ldq, qi, live && qval, kk, lane, hp.bias_u);
    qv[kk] = frag_global<T>((const T*)hp.q, a.ldq, qi, live && qval, kk, lane, hp.bias_v);
  }
  // blocks 0 .. FW - 1 of step 0 (block FW comes with fetch(0)): every load of the prologue is issued before the first LDS store
  {
    Vec16<T> rg[FW];
#pragma unroll
    for (int n = 0; n < FW; ++n) rg[n] = buf_load16<T>(rsP, p_lane + (unsigned)(32 * n) * pstride);
    fetch(0);
#pragma unroll
    for (int n = 0; n < FW; ++n) store16(ring + (n * 32 + trow) * LD + piece, rg[n]);
  }
  stash(0);
  fetch(1);
  __syncthreads();

  constexpr bool mbits = DM == 1;
  const __amdgpu_buffer_rsrc_t rsM = make_rsrc(a.keep_mask);
  const unsigned mrow = (unsigned)((((long)b * a.Tq + qi) * a.H + h) * a.keep_nw * 4);
  auto mask_word = [&](const int step) -> unsigned {
    return mbits ? __builtin_amdgcn_raw_buffer_load_b32(rsM, (live && qval && step < nstep) ? mrow + 4u * (unsigned)step : EMO_OOB, 0, 0) : 0u;
  };
  unsigned mw_next = mask_word(0);
  const float c2 = a.scale * 1.4426950408889634f;
  const unsigned keep_bits = __float_as_uint(a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f);
  float m = -INFINITY, l = 0.f;   // (m in the exp2 domain)
  f32x16 o[2];
  zero16(o[0]); zero16(o[1]);

  typename M_::Frag bcar[NK];   // fragments of this wave's lower band block of the coming step (block FW - 1 - wave + step)
  if (live) {
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) bcar[kk] = M_::load_kc(ring + ((FW - 1 - wave) % NRING) * 32 * LD, LD, 0, kk * M_::KSTEP, lane);
  }
  for (int step = 0; step < nstep; ++step) {
    const int j0 = step * 32;
    const unsigned mw = mw_next >> (4 * hh);   // this half wave's keys are 8 g + 4 hh + e
    mw_next = mask_word(step + 1);
    if (live) {
      // this wave's 64 band rows: blocks FW - 1 - wave + step and the next
      // (the upper block's fragments stay in registers: they are the next step's lower block -- one block read from LDS per step)
      const T* Bs1 = ring + ((FW - wave + step) % NRING) * 32 * LD;
      f32x16 s, g0, g1;
      zero16(s); zero16(g0); zero16(g1);
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        const typename M_::Frag bn = M_::load_kc(Bs1, LD, 0, kk * M_::KSTEP, lane);
        g0 = M_::mma(bcar[kk], qv[kk], g0);   // g[c][i]
        g1 = M_::mma(bn, qv[kk], g1);
        bcar[kk] = bn;
      }
      // (the score product is issued BEFORE the band tiles are stored: the stores wait for the band MFMAs' results while the matrix
      // pipe works on S)
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) s = M_::mma(M_::load_kc(Ks, LD, 0, kk * M_::KSTEP, lane), qu[kk], s);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {   // accumulator registers 4 q .. 4 q + 3 are band columns 8 q + 4 hh + 0 .. 3 of query il
        *reinterpret_cast<f32x4*>(Gs + il * LDG + 8 * q + 4 * hh) = f32x4{g0[4 * q], g0[4 * q + 1], g0[4 * q + 2], g0[4 * q + 3]};
        *reinterpret_cast<f32x4*>(Gs + il * LDG + 32 + 8 * q + 4 * hh) = f32x4{g1[4 * q], g1[4 * q + 1], g1[4 * q + 2], g1[4 * q + 3]};
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] += Gs[il * (LDG - 1) + 31 + c_row(r, lane)];   // G[il][31 - il + key]
      __builtin_amdgcn_wave_barrier();
      float mt = -INFINITY;
      if (j0 + 32 <= kend) {   // (wave-uniform) a tile without masked keys
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[r] *= c2;
          mt = fmaxf(mt, s[r]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[r] = (j0 + c_row(r, lane) >= kend) ? -INFINITY : s[r] * c2;
          mt = fmaxf(mt, s[r]);
        }
      }
      mt = xhalf_max(mt);
      const float mn = fmaxf(m, mt);
      const float mref = (mn == -INFINITY) ? 0.f : mn;
      const float alpha = __builtin_amdgcn_exp2f(m - mref);
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(s[r] - mref);
        rs += p;
        s[r] = p;
      }
      rs = xhalf_sum(rs);
      l = l * alpha + rs;
      m = mn;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
      if constexpr (DM != 0) {
        if constexpr (mbits) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            s[r] *= __uint_as_float(bit_of_row(mw, r) & keep_bits);
        } else {
          const float keep = 1.f / (1.f - a.drop_p);
          const uint32_t thr = dropout_thr(a.drop_p);
          const uint64_t dbase = drop_index(a, b, h, qi, j0 + 4 * hh);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const uint64_t pr = (dbase + (uint64_t)(8 * g)) >> 1;
            bool k0, k1, k2, k3;
            dropout_keep2(a.seed, pr, thr, k0, k1);
            dropout_keep2(a.seed, pr + 1, thr, k2, k3);
            s[4 * g] *= k0 ? keep : 0.f; s[4 * g + 1] *= k1 ? keep : 0.f;
            s[4 * g + 2] *= k2 ? keep : 0.f; s[4 * g + 3] *= k3 ? keep : 0.f;
          }
        }
      }
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int ks = 0; ks < NS; ++ks)
          o[dt] = M_::mma(chain_a<T, TR>(Vs, ks, 32 * dt, lane), chain_b<T>(s, ks), o[dt]);
    }
    lds_barrier();   // every wave has read the stage
    stash(step + 1);
    fetch(step + 2);
    lds_barrier();   // stage ready
  }
  if (!live) return;
  const float inv = l > 0.f ? 1.f / l : 0.f;
  store_dT<T>((T*)hp.out, a.ldo, i0, a.Tq, o, inv, lane);
  if (lane < 32 && qval) hp.lse[qi] = l > 0.f ? m * 0.6931471805599453f + __logf(l) : -INFINITY;
}

// delta[b,h,i] = sum_d dout[b,i,h,d] * out[b,i,h,d]; optionally also Q + pos_bias_u / Q + pos_bias_v as
// dense [B,Tq,H*DK] tensors (operands of the dK and dpos GEMMs).  8 lanes x 8 elements per (b,i,h) row.
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(const emoasr_attn_t a) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // over B*Tq*H rows x 8 lanes
  const long row = idx >> 3, total = (long)a.B * a.Tq * a.H;
  const int d0 = (int)(idx & 7) * 8;
  const bool ok = row < total;
  const int h = ok ? (int)(row % a.H) : 0;
  const long bt = ok ? row / a.H : 0;
  float dv[8], ov[8];
  load8<T>((const T*)a.dout + bt * a.ldo + h * DK + d0, dv);
  load8<T>((const T*)a.out + bt * a.ldo + h * DK + d0, ov);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += dv[j] * ov[j];
  s = sum8(s);
  if (!ok) return;
  const int i = (int)(bt % a.Tq), b = (int)(bt / a.Tq);
  if (d0 == 0) a.delta[((long)b * a.H + h) * a.Tq + i] = s;
  if (a.qu) {
    float qx[8], u[8], v[8];
    load8<T>((const T*)a.q + bt * a.ldq + h * DK + d0, qx);
#pragma unroll
    for (int j = 0; j < 8; ++j) { u[j] = qx[j] + a.bias_u[h * DK + d0 + j]; v[j] = qx[j] + a.bias_v[h * DK + d0 + j]; }
    const long o = (bt * a.H + h) * DK + d0;
    store8<T>((T*)a.qu + o, u);
    store8<T>((T*)a.qv + o, v);
  }
}

// dbias_u / dbias_v += sum over (batch, query tile) of the partials the dq kernel wrote:
// part[blk][h][which][d]; one block per (h, which), 4 waves split the partial rows
__global__ __launch_bounds__(256) void attn_dbias_reduce_kernel(int nblk, int H, const float* __restrict__ part,
                                                                float* __restrict__ dbias_u,
                                                                float* __restrict__ dbias_v) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = blockIdx.x >> 1, which = blockIdx.x & 1;
  float* dst = which ? dbias_v : dbias_u;
  if (!dst) return;
  float s = 0.f;
#pragma unroll 8
  for (int k = wave; k < nblk; k += 4) s += part[(((long)k * H + h) * 2 + which) * DK + lane];
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0) dst[h * DK + lane] += red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}

// ====================================================================================
// backward: dq (+ dbias_u, dbias_v).  Swapped tiles like the forward.
// ====================================================================================
constexpr int DQ_GS_FLOATS = 2176;  // >= 64*32 (score skew tile) and 2*32*33 (dS / P images)

template <typename T>
__device__ __forceinline__ void store_vec8(T* p, const float (&o)[8]) {
  if constexpr (sizeof(T) == 2) {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16)o[e];
    *reinterpret_cast<bf16x8*>(p) = v;
  } else {
    *reinterpret_cast<f32x4*>(p) = f32x4{o[0], o[1], o[2], o[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{o[4], o[5], o[6], o[7]};
  }
}

// STORED: the forward kept the scaled scores S^T (a.st); no score recomputation here and the
// (q+v) part of dq is left to a batched GEMM over the dBD band this kernel stores.
template <typename T, bool TR, bool STORED>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const emoasr_attn_t a) {
  using M_ = Mma<T>;
  constexpr int NK = AttnCfg<T>::NK, NS = AttnCfg<T>::NS, LD = AttnCfg<T>::LD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;  // one query tile per block; the 4 waves split the key tiles
  if (i0 >= a.Tq) return;
  constexpr int PD_OFF = 32 * 33;              // second [32][33] f32 tile (dropped probabilities)
  constexpr int WAVE_BYTES = DQ_GS_FLOATS * 4 + 96 * LD * (int)sizeof(T);
  float* Gs = reinterpret_cast<float*>(smem + wave * WAVE_BYTES);
  T* Ks = reinterpret_cast<T*>(smem + wave * WAVE_BYTES + DQ_GS_FLOATS * 4);  // [32][LD]
  T* Bs = Ks + 32 * LD;                                                        // [64][LD] band rows

  const HeadPtrs hp = head_ptrs<T>(a, b, h);
  const int qi = i0 + (lane & 31);
  const bool qval = qi < a.Tq;
  typename M_::Frag qu[STORED ? 1 : NK], qv[STORED ? 1 : NK], dof[NK];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    if constexpr (!STORED) {
      qu[kk] = frag_global<T>((const T*)hp.q, a.ldq, qi, qval, kk, lane, hp.bias_u);
      qv[kk] = hp.pos ? frag_global<T>((const T*)hp.q, a.ldq, qi, qval, kk, lane, hp.bias_v) : qu[kk];
    }
    dof[kk] = frag_global<T>((const T*)hp.dout, a.ldo, qi, qval, kk, lane, nullptr);
  }
  const float lse_q = qval ? hp.lse[qi] : -INFINITY;
  const float del_q = qval ? hp.delta[qi] : 0.f;
  const uint64_t drop_base = drop_index(a, b, h, qi, 0);
  f32x16 dqu[2], dqv[2];
  zero16(dqu[0]); zero16(dqu[1]); zero16(dqv[0]); zero16(dqv[1]);
  int kend = hp.klen;
  if (a.causal) kend = min(kend, i0 + 32);
  for (int j0 = wave * 32; j0 < kend; j0 += 128) {
    f32x16 s;
    if constexpr (STORED) {
      const float* srow = a.st + (((long)b * a.H + h) * a.Tk + j0) * a.ldst + qi;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kl = c_row(r, lane);
        s[r] = (qval && j0 + kl < a.Tk) ? srow[(long)kl * a.ldst] : 0.f;
      }
    } else {
      score_tile<T, true>(s, a, hp, i0, j0, qu, qv, nullptr, Gs, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] *= a.scale;
    }
    // dP^T = V . dO^T   (rows keys, cols queries)
    f32x16 dp;
    zero16(dp);
    {
      const int krow = j0 + (lane & 31);
      const bool kval = krow < a.Tk;
#pragma unroll
      for (int kk = 0; kk < NK; ++kk)
        dp = M_::mma(frag_global<T>((const T*)hp.v, a.ldv, krow, kval, kk, lane, nullptr), dof[kk], dp);
    }
    f32x16 ds;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kj = j0 + c_row(r, lane);
      const bool masked = kj >= hp.klen || (a.causal && kj > qi) || lse_q == -INFINITY;
      const float p = masked ? 0.f : __expf(s[r] - lse_q);
      float dpr = dp[r];
      float dsc = 1.f;
      if (a.drop_p > 0.f) dsc = dropout_scale(a.seed, drop_base + (uint64_t)kj, a.drop_p);
      dpr *= dsc;
      ds[r] = p * (dpr - del_q) * a.scale;
      // tiles go through wave-private LDS ([key][query], stride 33): the same image feeds the
      // 16-byte row stores below and the band un-skew further down
      Gs[c_row(r, lane) * 33 + (lane & 31)] = ds[r];
      Gs[PD_OFF + c_row(r, lane) * 33 + (lane & 31)] = p * dsc;
    }
    // dQu^T += K^T . dS^T
    stage_rows<T, 32>(Ks, (const T*)hp.k, a.ldk, j0, 0, a.Tk, lane, nullptr);
    __builtin_amdgcn_wave_barrier();
    if (a.pdT) {
      // materialised mode: P^T (after dropout) and dS^T go to HBM once, as whole 16-query row pieces;
      // dV, dK and dpos are then (batched) GEMMs over them instead of more score recomputations.
      const int kl = lane >> 1, c0 = (lane & 1) * 16;
      if (j0 + kl < a.Tk) {
        const long o = (((long)b * a.H + h) * a.Tk + j0 + kl) * a.ldpd + i0 + c0;
#pragma unroll
        for (int g8 = 0; g8 < 2; ++g8) {
          if (i0 + c0 + 8 * g8 < a.ldpd) {  // padded columns may hold anything (never multiplied in)
            float vd[8], vp[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              vd[e] = Gs[kl * 33 + c0 + 8 * g8 + e];
              vp[e] = Gs[PD_OFF + kl * 33 + c0 + 8 * g8 + e];
            }
            store_vec8<T>((T*)a.dsT + o + 8 * g8, vd);
            store_vec8<T>((T*)a.pdT + o + 8 * g8, vp);
          }
        }
      }
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int ks = 0; ks < NS; ++ks)
        dqu[dt] = M_::mma(chain_a<T, TR>(Ks, ks, 32 * dt, lane), chain_b<T>(ds, ks), dqu[dt]);
    if (hp.pos) {
      // un-skew dS^T (already in Gs, [key][query] stride 33) into the band: dG^T[c][i] = dS^T[c - 31 + i][i]
      const int rbase = a.Tq - 32 - i0 + j0;
      if constexpr (!STORED) stage_rows<T, 64>(Bs, (const T*)hp.pos, a.ldp, rbase, 0, 2 * a.Tq - 1, lane, nullptr);
      __builtin_amdgcn_wave_barrier();
      if (a.dbd) {
        // dBD[h, b, i, r] (r = table row, contiguous): lane <-> band column c, one query row per
        // iteration, so every store is a contiguous 64-wide row segment.  Each (i, r) pair belongs
        // to exactly one key tile, so this is a plain store.
        T* drow = (T*)a.dbd + (((long)h * a.B + b) * a.Tq + i0) * a.ldbd;
        const int row = rbase + lane;
        const bool rok = row >= 0 && row < 2 * a.Tq - 1;
        for (int il = 0; il < 32; ++il) {
          const int key = lane - 31 + il;
          if (rok && key >= 0 && key < 32 && i0 + il < a.Tq)
            drow[(long)il * a.ldbd + row] = from_f32<T>(Gs[key * 33 + il]);
        }
      }
      if constexpr (!STORED)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        f32x16 dg;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = 32 * ct + c_row(r, lane) - 31 + (lane & 31);
          dg[r] = (key >= 0 && key < 32) ? Gs[key * 33 + (lane & 31)] : 0.f;
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int ks = 0; ks < NS; ++ks)
            dqv[dt] = M_::mma(chain_a<T, TR>(Bs + 32 * ct * LD, ks, 32 * dt, lane), chain_b<T>(dg, ks), dqv[dt]);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  // fold the four waves' partial dQu / dQv (each saw a quarter of the key tiles) through LDS; the
  // wave-private staging regions are free now
  {
    float* mine = reinterpret_cast<float*>(smem + wave * WAVE_BYTES);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        mine[((dt * 16 + r) * 2 + 0) * 64 + lane] = dqu[dt][r];
        mine[((dt * 16 + r) * 2 + 1) * 64 + lane] = dqv[dt][r];
      }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float* other = reinterpret_cast<const float*>(smem + w * WAVE_BYTES);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          dqu[dt][r] += other[((dt * 16 + r) * 2 + 0) * 64 + lane];
          dqv[dt][r] += other[((dt * 16 + r) * 2 + 1) * 64 + lane];
        }
    }
  }
  // bias gradients: sum over the 32 queries of this tile (lanes within each half)
  if (a.dbias_u || a.dbias_v) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float su = qval ? dqu[dt][r] : 0.f, sv = qval ? dqv[dt][r] : 0.f;
        su = half_sum32(su);
        sv = half_sum32(sv);
        if ((lane & 31) == 31) {
          const int d = h * DK + 32 * dt + c_row(r, lane);
          if (a.dbias_u) atomicAdd(&a.dbias_u[d], su);
          if (!STORED && a.dbias_v && hp.pos) atomicAdd(&a.dbias_v[d], sv);
        }
      }
  }
  if (!STORED && hp.pos) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dqu[dt][r] += dqv[dt][r];
  }
  store_dT<T>((T*)hp.dq, a.ldq, i0, a.Tq, dqu, 1.f, lane);
}

// ====================================================================================
// backward, materialised mode (the default path): dq, dbias_u, dbias_v and the P^T / dS^T / dBD
// images the batched GEMMs of launch_bwd_tr consume.  Same mathematics as attn_bwd_dq_kernel,
// restructured around memory latency:
//   * every global operand of a key tile (K rows, the 64 band rows of pos, the V fragments) is
//     fetched ONCE, one tile ahead, into registers with bounds-checked buffer loads while the
//     current tile is being worked on; K and pos reach the MFMAs through LDS in both roles
//     (score operand via k-contiguous reads, dQ operand via transposed reads);
//   * the band skew runs in two 32x32 halves and the dS / P images are kept in T, so a wave needs
//     ~18.5 KB of LDS in bf16: two blocks (8 waves) per CU instead of one.
// ====================================================================================
#ifndef EMO_DQ_SKIP
#define EMO_DQ_SKIP 0  // timing experiments only: bit mask of pieces to leave out (results are then wrong)
#endif
template <typename T> struct DqCfg {
  static constexpr int IMG_LD = sizeof(T) == 2 ? 40 : 33;  // image row stride (bf16: 80 B rows -> 16-byte reads)
  static constexpr int IMG_ELEMS = 32 * IMG_LD;
  static constexpr int IMG_BYTES = 2 * IMG_ELEMS * (int)sizeof(T);
  static constexpr int GS_BYTES = ((IMG_BYTES > 32 * 32 * 4 ? IMG_BYTES : 32 * 32 * 4) + 15) / 16 * 16;
  static constexpr int RED_BYTES = 2 * 32 * 64 * 4;         // cross-wave dq reduction image
  static constexpr int STAGE_BYTES = 96 * AttnCfg<T>::LD * (int)sizeof(T);
  static constexpr int WAVE_BYTES =
      GS_BYTES + STAGE_BYTES > RED_BYTES ? GS_BYTES + STAGE_BYTES : RED_BYTES;
  static constexpr int KR = 32 * DK * (int)sizeof(T) / 16 / 64;  // 16-byte pieces of a 32-row tile per lane
};

template <typename T, bool TR>
__global__ __launch_bounds__(256) void attn_bwd_dq2_kernel(const emoasr_attn_t a) {
  using M_ = Mma<T>;
  using C_ = DqCfg<T>;
  constexpr int NK = AttnCfg<T>::NK, NS = AttnCfg<T>::NS, LD = AttnCfg<T>::LD;
  constexpr int KR = C_::KR, IMG = C_::IMG_LD, VEC = 16 / sizeof(T), PER_ROW = DK / VEC;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, il = lane & 31;
  const int i0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;
  if (i0 >= a.Tq) return;
  char* mine = smem + wave * C_::WAVE_BYTES;
  float* Gs = reinterpret_cast<float*>(mine);              // one 32x32 f32 skew half ...
  T* img_ds = reinterpret_cast<T*>(mine);                  // ... later reused by the dS^T / P^T images
  T* img_p = img_ds + C_::IMG_ELEMS;
  T* Ks = reinterpret_cast<T*>(mine + C_::GS_BYTES);       // [32][LD]
  T* Bs = Ks + 32 * LD;                                    // [64][LD] band rows

  const HeadPtrs hp = head_ptrs<T>(a, b, h);
  const bool rel = hp.pos != nullptr;
  const int qi = i0 + il;
  const bool qval = qi < a.Tq;
  typename M_::Frag qu[NK], qv[NK], dof[NK];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    qu[kk] = frag_global<T>((const T*)hp.q, a.ldq, qi, qval, kk, lane, hp.bias_u);
    qv[kk] = rel ? frag_global<T>((const T*)hp.q, a.ldq, qi, qval, kk, lane, hp.bias_v) : qu[kk];
    dof[kk] = frag_global<T>((const T*)hp.dout, a.ldo, qi, qval, kk, lane, nullptr);
  }
  const float lse_q = qval ? hp.lse[qi] : -INFINITY;
  const float del_q = qval ? hp.delta[qi] : 0.f;
  const uint64_t drop_base = drop_index(a, b, h, qi, 0);
  f32x16 dqu[2], dqv[2];
  zero16(dqu[0]); zero16(dqu[1]); zero16(dqv[0]); zero16(dqv[1]);
  int kend = hp.klen;
  if (a.causal) kend = min(kend, i0 + 32);

  // ---- one-tile-ahead operand fetch ------------------------------------------------------
  const __amdgpu_buffer_rsrc_t rsK = make_rsrc(hp.k), rsP = make_rsrc(rel ? hp.pos : hp.k);
  Vec16<T> kreg[KR], preg[2 * KR];
  typename M_::Frag vf[NK];
  auto fetch = [&](int j0) {
    const bool live = j0 < kend;  // past the last tile: every offset out of bounds, no traffic
#pragma unroll
    for (int i = 0; i < KR; ++i) {
      const int v = lane + 64 * i, r = v / PER_ROW, piece = (v % PER_ROW) * VEC, row = j0 + r;
      kreg[i] = buf_load16<T>(rsK, live && row < a.Tk ? (unsigned)(((long)row * a.ldk + piece) * sizeof(T)) : EMO_OOB);
    }
    if (rel) {
      const int rbase = a.Tq - 32 - i0 + j0;
#pragma unroll
      for (int i = 0; i < 2 * KR; ++i) {
        const int v = lane + 64 * i, r = v / PER_ROW, piece = (v % PER_ROW) * VEC, row = rbase + r;
        const bool ok = live && row >= 0 && row < 2 * a.Tq - 1;
        preg[i] = buf_load16<T>(rsP, ok ? (unsigned)(((long)row * a.ldp + piece) * sizeof(T)) : EMO_OOB);
      }
    }
    const int krow = j0 + il;
#pragma unroll
    for (int kk = 0; kk < NK; ++kk)
      vf[kk] = frag_global<T>((const T*)hp.v, a.ldv, krow, live && krow < a.Tk, kk, lane, nullptr);
  };
  fetch(wave * 32);

  for (int j0 = wave * 32; j0 < kend; j0 += 128) {
    // registers -> LDS (the previous iteration ended with a wave barrier: the stage is free)
#pragma unroll
    for (int i = 0; i < KR; ++i) {
      const int v = lane + 64 * i;
      lds_stage16(Ks + (v / PER_ROW) * LD + (v % PER_ROW) * VEC, kreg[i]);
    }
    if (rel) {
#pragma unroll
      for (int i = 0; i < 2 * KR; ++i) {
        const int v = lane + 64 * i;
        lds_stage16(Bs + (v / PER_ROW) * LD + (v % PER_ROW) * VEC, preg[i]);
      }
    }
    // dP^T = V . dO^T   (rows keys, cols queries), then the next tile's loads go out
    f32x16 dp;
    zero16(dp);
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) dp = M_::mma(vf[kk], dof[kk], dp);
    __builtin_amdgcn_wave_barrier();
    fetch(j0 + 128);

    // S^T = K . (Q+u)^T + skew(pos_band . (Q+v)^T)
    f32x16 s;
    zero16(s);
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) s = M_::mma(M_::load_kc(Ks, LD, 0, kk * M_::KSTEP, lane), qu[kk], s);
    if (rel && !(EMO_DQ_SKIP & 8)) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        f32x16 g;
        zero16(g);
#pragma unroll
        for (int kk = 0; kk < NK; ++kk)
          g = M_::mma(M_::load_kc(Bs + 32 * ct * LD, LD, 0, kk * M_::KSTEP, lane), qv[kk], g);  // g[c][i]
#pragma unroll
        for (int r = 0; r < 16; ++r) Gs[c_row(r, lane) * 32 + il] = g[r];
        __builtin_amdgcn_wave_barrier();
        // element (key jl, query il) sits in band column c = 31 - il + jl: first half iff jl <= il
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int jl = c_row(r, lane);
          const int c = ct == 0 ? 31 - il + jl : jl - il - 1;
          const bool use = ct == 0 ? jl <= il : jl > il;
          const float gv = Gs[(use ? c : 0) * 32 + il];
          s[r] += use ? gv : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    f32x16 ds;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kj = j0 + c_row(r, lane);
      const bool masked = kj >= hp.klen || (a.causal && kj > qi) || lse_q == -INFINITY;
      const float p = masked ? 0.f : __expf(s[r] * a.scale - lse_q);
      float dsc = 1.f;
      if (a.drop_p > 0.f && !(EMO_DQ_SKIP & 4)) dsc = dropout_scale(a.seed, drop_base + (uint64_t)kj, a.drop_p);
      ds[r] = p * (dp[r] * dsc - del_q) * a.scale;
      img_ds[c_row(r, lane) * IMG + il] = from_f32<T>(ds[r]);
      img_p[c_row(r, lane) * IMG + il] = from_f32<T>(p * dsc);
    }
    __builtin_amdgcn_wave_barrier();
    {
      // P^T (after dropout) and dS^T go to HBM once, as 16-query row pieces
      const int kl = lane >> 1, c0 = (lane & 1) * 16;
      if (j0 + kl < a.Tk && !(EMO_DQ_SKIP & 2)) {
        const long o = (((long)b * a.H + h) * a.Tk + j0 + kl) * a.ldpd + i0 + c0;
#pragma unroll
        for (int g8 = 0; g8 < 2; ++g8) {
          if (i0 + c0 + 8 * g8 < a.ldpd) {  // padded columns may hold anything (never multiplied in)
            if constexpr (sizeof(T) == 2) {
              *reinterpret_cast<bf16x8*>((T*)a.dsT + o + 8 * g8) =
                  *reinterpret_cast<const bf16x8*>(img_ds + kl * IMG + c0 + 8 * g8);
              *reinterpret_cast<bf16x8*>((T*)a.pdT + o + 8 * g8) =
                  *reinterpret_cast<const bf16x8*>(img_p + kl * IMG + c0 + 8 * g8);
            } else {
              float vd[8], vp[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                vd[e] = to_f32(img_ds[kl * IMG + c0 + 8 * g8 + e]);
                vp[e] = to_f32(img_p[kl * IMG + c0 + 8 * g8 + e]);
              }
              store_vec8<T>((T*)a.dsT + o + 8 * g8, vd);
              store_vec8<T>((T*)a.pdT + o + 8 * g8, vp);
            }
          }
        }
      }
    }
    // dQu^T += K^T . dS^T
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int ks = 0; ks < NS; ++ks)
        dqu[dt] = M_::mma(chain_a<T, TR>(Ks, ks, 32 * dt, lane), chain_b<T>(ds, ks), dqu[dt]);
    if (rel) {
      const int rbase = a.Tq - 32 - i0 + j0;
      if (a.dbd && !(EMO_DQ_SKIP & 1)) {
        // dBD[h, b, i, r] (r = table row, contiguous): lane <-> band column, one query row per
        // iteration -> contiguous 64-wide row segments.  Each (i, r) belongs to exactly one key tile.
        // element stores through a buffer descriptor: one 32-bit offset per store, out-of-range lanes are
        // dropped by the bounds check instead of being branched around (32 stores per tile)
        const __amdgpu_buffer_rsrc_t rsBd = make_rsrc((T*)a.dbd + (((long)h * a.B + b) * a.Tq + i0) * a.ldbd);
        const int row = rbase + lane;
        const bool rok = row >= 0 && row < 2 * a.Tq - 1;
        const int qmax = min(32, a.Tq - i0);
#pragma unroll 8
        for (int q = 0; q < 32; ++q) {
          const int key = lane - 31 + q;
          const bool ok = rok && key >= 0 && key < 32 && q < qmax;
          buf_store_elem<T>(rsBd, ok ? (unsigned)(((long)q * a.ldbd + row) * sizeof(T)) : EMO_OOB,
                            to_f32(img_ds[(ok ? key : 0) * IMG + q]));
        }
      }
      if (!(EMO_DQ_SKIP & 16))
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        f32x16 dg;  // dG^T[c][i] = dS^T[c - 31 + i][i]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = 32 * ct + c_row(r, lane) - 31 + il;
          const bool in = key >= 0 && key < 32;
          const float gv = to_f32(img_ds[(in ? key : 0) * IMG + il]);
          dg[r] = in ? gv : 0.f;
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int ks = 0; ks < NS; ++ks)
            dqv[dt] = M_::mma(chain_a<T, TR>(Bs + 32 * ct * LD, ks, 32 * dt, lane), chain_b<T>(dg, ks), dqv[dt]);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  // fold the four waves' partial dQu / dQv through LDS (the wave-private regions are free now)
  {
    float* red = reinterpret_cast<float*>(mine);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        red[((dt * 16 + r) * 2 + 0) * 64 + lane] = dqu[dt][r];
        red[((dt * 16 + r) * 2 + 1) * 64 + lane] = dqv[dt][r];
      }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float* other = reinterpret_cast<const float*>(smem + w * C_::WAVE_BYTES);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          dqu[dt][r] += other[((dt * 16 + r) * 2 + 0) * 64 + lane];
          dqv[dt][r] += other[((dt * 16 + r) * 2 + 1) * 64 + lane];
        }
    }
  }
  if (a.dbias_u || a.dbias_v) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float su = qval ? dqu[dt][r] : 0.f, sv = qval ? dqv[dt][r] : 0.f;
        su = half_sum32(su);
        sv = half_sum32(sv);
        if (il == 31) {
          const int d = 32 * dt + c_row(r, lane);
          if (a.dbias_part) {  // per-(batch, query tile) partials, folded by attn_dbias_reduce_kernel
            float* pp = a.dbias_part + ((((long)b * gridDim.x + blockIdx.x) * a.H + h) * 2) * DK + d;
            pp[0] = su;
            pp[DK] = rel ? sv : 0.f;
          } else {
            if (a.dbias_u) atomicAdd(&a.dbias_u[h * DK + d], su);
            if (a.dbias_v && rel) atomicAdd(&a.dbias_v[h * DK + d], sv);
          }
        }
      }
  }
  if (rel) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dqu[dt][r] += dqv[dt][r];
  }
  store_dT<T>((T*)hp.dq, a.ldq, i0, a.Tq, dqu, 1.f, lane);
}

// ====================================================================================
// backward: dk, dv.  Non-swapped tiles (rows = queries, cols = this wave's 32 keys).
// ====================================================================================
template <typename T, bool TR>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const emoasr_attn_t a) {
  using M_ = Mma<T>;
  constexpr int NK = AttnCfg<T>::NK, NS = AttnCfg<T>::NS, LD = AttnCfg<T>::LD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j0 = (blockIdx.x * 4 + wave) * 32, h = blockIdx.y, b = blockIdx.z;
  if (j0 >= a.Tk) return;
  constexpr int WAVE_BYTES = 32 * 64 * 4 + 64 * LD * (int)sizeof(T);
  float* Gs = reinterpret_cast<float*>(smem + wave * WAVE_BYTES);
  T* Qs = reinterpret_cast<T*>(smem + wave * WAVE_BYTES + 32 * 64 * 4);  // [32][LD] q + u
  T* Os = Qs + 32 * LD;                                                   // [32][LD] dO

  const HeadPtrs hp = head_ptrs<T>(a, b, h);
  const int kj = j0 + (lane & 31);
  const bool kval = kj < a.Tk;
  typename M_::Frag kf[NK], vf[NK];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    kf[kk] = frag_global<T>((const T*)hp.k, a.ldk, kj, kval, kk, lane, nullptr);
    vf[kk] = frag_global<T>((const T*)hp.v, a.ldv, kj, kval, kk, lane, nullptr);
  }
  f32x16 dk[2], dv[2];
  zero16(dk[0]); zero16(dk[1]); zero16(dv[0]); zero16(dv[1]);
  const bool key_dead = kj >= hp.klen;
  const int istart = a.causal ? (j0 / 32) * 32 : 0;
  if (j0 < hp.klen) {
    for (int i0 = istart; i0 < a.Tq; i0 += 32) {
      const int qrow = i0 + (lane & 31);
      const bool qrv = qrow < a.Tq;
      typename M_::Frag qu[NK], qv[NK], dof[NK];
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        qu[kk] = frag_global<T>((const T*)hp.q, a.ldq, qrow, qrv, kk, lane, hp.bias_u);
        qv[kk] = hp.pos ? frag_global<T>((const T*)hp.q, a.ldq, qrow, qrv, kk, lane, hp.bias_v) : qu[kk];
        dof[kk] = frag_global<T>((const T*)hp.dout, a.ldo, qrow, qrv, kk, lane, nullptr);
      }
      f32x16 s;
      score_tile<T, false>(s, a, hp, i0, j0, qu, qv, kf, Gs, lane);
      f32x16 dp;
      zero16(dp);
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) dp = M_::mma(dof[kk], vf[kk], dp);  // dO . V^T
      stage_rows<T, 32>(Qs, (const T*)hp.q, a.ldq, i0, 0, a.Tq, lane, hp.bias_u);
      stage_rows<T, 32>(Os, (const T*)hp.dout, a.ldo, i0, 0, a.Tq, lane, nullptr);
      f32x16 pd, ds;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = i0 + c_row(r, lane);
        const bool qv_ = qi < a.Tq;
        const float lse_q = qv_ ? hp.lse[qi] : -INFINITY;
        const float del_q = qv_ ? hp.delta[qi] : 0.f;
        const bool masked = key_dead || (a.causal && kj > qi) || lse_q == -INFINITY;
        const float p = masked ? 0.f : __expf(s[r] * a.scale - lse_q);
        const float dsc = a.drop_p > 0.f ? dropout_scale(a.seed, drop_index(a, b, h, qi, kj), a.drop_p) : 1.f;
        pd[r] = p * dsc;
        ds[r] = p * (dp[r] * dsc - del_q) * a.scale;
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int ks = 0; ks < NS; ++ks) {
          dv[dt] = M_::mma(chain_a<T, TR>(Os, ks, 32 * dt, lane), chain_b<T>(pd, ks), dv[dt]);
          dk[dt] = M_::mma(chain_a<T, TR>(Qs, ks, 32 * dt, lane), chain_b<T>(ds, ks), dk[dt]);
        }
      __builtin_amdgcn_wave_barrier();
    }
  }
  store_dT<T>((T*)hp.dk, a.ldk, j0, a.Tk, dk, 1.f, lane);
  store_dT<T>((T*)hp.dv, a.ldv, j0, a.Tk, dv, 1.f, lane);
}

// ====================================================================================
// backward: dpos[r, h*64 + d] += sum_{b,i} dBD[b,h,i, j = i - rel(r)] * (q[b,i,h,:] + v)[d]
// block = (32 table rows, h, b); the four waves take query tiles round-robin and are
// reduced through LDS before one set of f32 atomics.
// ====================================================================================
template <typename T, bool TR>
__global__ __launch_bounds__(256) void attn_bwd_dpos_kernel(const emoasr_attn_t a) {
  using M_ = Mma<T>;
  constexpr int NK = AttnCfg<T>::NK, NS = AttnCfg<T>::NS, LD = AttnCfg<T>::LD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;
  constexpr int WAVE_BYTES = 2 * 32 * 64 * 4 + 32 * LD * (int)sizeof(T);
  float* Gs = reinterpret_cast<float*>(smem + wave * WAVE_BYTES);
  float* Es = Gs + 32 * 64;                                                   // [32][64]
  T* Qs = reinterpret_cast<T*>(smem + wave * WAVE_BYTES + 2 * 32 * 64 * 4);  // [32][LD] q + v

  const HeadPtrs hp = head_ptrs<T>(a, b, h);
  f32x16 acc[2];
  zero16(acc[0]); zero16(acc[1]);
  for (int i0 = wave * 32; i0 < a.Tq; i0 += 128) {
    const int jbase = i0 + r0 - a.Tq + 1;  // key of (i_l = 0, r_l = 0); keys jbase .. jbase+62
    if (jbase + 62 < 0 || jbase >= hp.klen) continue;
    const int qrow = i0 + (lane & 31);
    const bool qrv = qrow < a.Tq;
    typename M_::Frag qu[NK], qv[NK], dof[NK];
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      qu[kk] = frag_global<T>((const T*)hp.q, a.ldq, qrow, qrv, kk, lane, hp.bias_u);
      qv[kk] = frag_global<T>((const T*)hp.q, a.ldq, qrow, qrv, kk, lane, hp.bias_v);
      dof[kk] = frag_global<T>((const T*)hp.dout, a.ldo, qrow, qrv, kk, lane, nullptr);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int j0 = jbase + 32 * t;
      const int kj = j0 + (lane & 31);
      const bool kval = kj >= 0 && kj < a.Tk;
      typename M_::Frag kf[NK], vf[NK];
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        kf[kk] = frag_global<T>((const T*)hp.k, a.ldk, kj, kval, kk, lane, nullptr);
        vf[kk] = frag_global<T>((const T*)hp.v, a.ldv, kj, kval, kk, lane, nullptr);
      }
      f32x16 s;
      score_tile<T, false>(s, a, hp, i0, j0, qu, qv, kf, Gs, lane);
      f32x16 dp;
      zero16(dp);
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) dp = M_::mma(dof[kk], vf[kk], dp);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = i0 + c_row(r, lane);
        const bool qv_ = qi < a.Tq;
        const float lse_q = qv_ ? hp.lse[qi] : -INFINITY;
        const float del_q = qv_ ? hp.delta[qi] : 0.f;
        const bool masked = kj < 0 || kj >= hp.klen || lse_q == -INFINITY;
        const float p = masked ? 0.f : __expf(s[r] * a.scale - lse_q);
        const float dsc = a.drop_p > 0.f && !masked ? dropout_scale(a.seed, drop_index(a, b, h, qi, kj), a.drop_p) : 1.f;
        Es[c_row(r, lane) * 64 + 32 * t + (lane & 31)] = p * (dp[r] * dsc - del_q) * a.scale;
      }
    }
    stage_rows<T, 32>(Qs, (const T*)hp.q, a.ldq, i0, 0, a.Tq, lane, hp.bias_v);
    __builtin_amdgcn_wave_barrier();
    f32x16 e;  // e[i_l][r_l] = dS[i, key - jbase = i_l + r_l]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int il = c_row(r, lane);
      e[r] = Es[il * 64 + il + (lane & 31)];
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int ks = 0; ks < NS; ++ks)
        acc[dt] = M_::mma(chain_a<T, TR>(Qs, ks, 32 * dt, lane), chain_b<T>(e, ks), acc[dt]);
    __builtin_amdgcn_wave_barrier();
  }
  // cross-wave reduction: reuse the first 32 KiB of LDS as [4][2][16][64] f32
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((wave * 2 + dt) * 16 + r) * 64 + lane] = acc[dt][r];
  __syncthreads();
  if (wave == 0) {
    const int row = r0 + (lane & 31);
    if (row <= 2 * a.Tq - 2) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) v += red[((w * 2 + dt) * 16 + r) * 64 + lane];
          atomicAdd(&a.dpos[(long)row * (a.H * DK) + h * DK + 32 * dt + c_row(r, lane)], v);
        }
    }
  }
}

// dk[b,key,h,:] += rowsum_q(dS^T[b,h,key,:]) * u[h,:]   (K^T-side of the (q+u) bias; one wave per key row)
template <typename T>
__global__ __launch_bounds__(256) void attn_dk_bias_kernel(const emoasr_attn_t a) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);  // over B*H*Tk
  if (row >= (long)a.B * a.H * a.Tk) return;
  const int key = row % a.Tk;
  const long bh = row / a.Tk;
  const int h = bh % a.H, b = bh / a.H;
  const T* src = (const T*)a.dsT + row * a.ldpd;
  float s = 0.f;
  for (int i = lane; i < a.Tq; i += 64) s += to_f32(src[i]);
  s = wave_sum(s);
  T* dk = (T*)a.dk + ((long)b * a.Tk + key) * a.ldk + h * DK + lane;
  *dk = from_f32<T>(to_f32(*dk) + s * a.bias_u[h * DK + lane]);
}
// dpos[r, h*64+d] += cs[h, r] * v[h*64+d]
__global__ __launch_bounds__(256) void attn_dpos_bias_kernel(const emoasr_attn_t a) {
  const int n = (2 * a.Tq - 1) * a.H * DK;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int col = i % (a.H * DK), r = i / (a.H * DK);
  a.dpos[i] += a.cs[(long)(col / DK) * a.ldbd + r] * a.bias_v[col];
}

// dbias_v[h*64+d] += sum_r cs[h, r] * pos[r, h*64+d]   (cs = column sums of dBD_h)
template <typename T>
__global__ __launch_bounds__(256) void attn_dbias_v_kernel(const emoasr_attn_t a) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= a.H * DK) return;
  const float* cs = a.cs + (long)(col / DK) * a.ldbd;
  const T* pos = (const T*)a.pos + col;
  float s = 0.f;
  for (int r = 0; r < 2 * a.Tq - 1; ++r) s += cs[r] * to_f32(pos[(long)r * a.ldp]);
  a.dbias_v[col] += s;
}

// ====================================================================================
// backward, single pass (round 2; bf16): one launch recomputes each score tile ONCE and produces dQ, dK, dV,
// dbias_u and -- with relative positions -- the dS image the dpos pass reads.  Nothing else reaches HBM: no
// P^T / dBD images, no follow-up GEMMs.
//
//   workgroup = FW waves = FW consecutive 32-key tiles of one (batch, head); it sweeps the query tiles i0 = 0, 32, ...
//   * per wave, stationary in registers: K and V fragments of its key tile, K^T fragments (dQ operand), the dK^T / dV^T
//     accumulators, per-key column sums of dS (dbias_u = sum_j colsum_j K_j);
//   * per step, shared by the block through LDS (double buffered, ONE barrier per step): the (Q+u), (Q+v), dO tiles
//     of the 32 queries and the union of the waves' position bands (32*FW + 32 rows of the projected table); they are
//     fetched one step ahead into registers with bounds-checked buffer loads and written to LDS at the end of the step;
//   * tile orientation as in the forward (rows = keys in registers, columns = queries on lanes): S^T = K (Q+u)^T +
//     skew(band (Q+v)^T), dP^T = V dO^T; dQ^T += K^T dS^T + band^T unskew(dS^T) chains the accumulator straight into
//     the next MFMA; dV^T += dO^T P and dK^T += (Q+u)^T dS sum over the query index, which sits on the lanes, so P and
//     dS go through a 2.5 KB wave-private bf16 image (the same image feeds the un-skew);
//   * each wave leaves its dQ^T partial in an LDS slab ([64 d][33] f32, in its skew region); after the barrier that ends
//     the step the block sums the slabs and stores ONE f32 partial per (key block, query row) with plain 256-byte row
//     stores (LDS float atomics measured ~170 cycles per wave-instruction, global float atomics would make dQ depend on
//     arrival order: bf16 rounding flips of dQ then propagate through every layer below); the finalize pass adds the key
//     blocks' partials in block order, so dQ is bit-reproducible;
//   * dS (query-major, bf16) is stored for attn_bwd_dpos3_kernel, which walks its diagonals: dpos[r] = sum_{b,i}
//     dS[b,i,i-(Tq-1)+r] (Q+v)[b,i].  dbias_v = colsum(dQ) - dbias_u: the flush also sums the dQ rows it stores.
// ====================================================================================

struct FusedWs {       // workspace carved by emoasr_attn_bwd_fused
  float* dq32;         // f32 [key blocks][B, Tq, H*DK]: one dQ partial per key block, summed in block order by the finalize pass
  long dq_slab;        // floats per key block
  void* dsq;           // T [B, H, Tq, ldds]: dS, query-major (relative positions only)
  long ldds;
  const void *qu, *qv; // T [B, Tq, ldqu]: Q + pos_bias_u, Q + pos_bias_v  (without biases: q itself, ldq)
  long ldqu;
  unsigned long long* stamp;  // -DEMO_ATTN_STAMP builds only: s_memtime of wave 0 of workgroup (0,0,0) at 13 points per step
  // two-pass backward: the attention-dropout keep mask as bits, one 32-bit word per (row, head, 32-key tile) -- [rows, H, mask_nw],
  // bit k of word w = key 32 w + k kept (attn_dropmask_kernel; nullptr without dropout)
  const unsigned* mask;
  int mask_nw;
  int img_from_kv;     // R6: the key pass writes the dS image and the query pass (attn_bwd_q2_kernel) reads it instead of recomputing
};

// prologue: delta[b,h,i] = dO.O, dense Q+u / Q+v.  8 lanes x 8 elements per (b,i,h) row.
// zero / zn: optional f32 buffer cleared on the side (the position-table gradient that attn_bwd_dpos3_kernel accumulates into:
// saves the separate memset launch of the per-layer backward)
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const emoasr_attn_t a, T* __restrict__ qu, T* __restrict__ qv,
                                                            float* __restrict__ zero, const long zn, const long nrows) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  for (long i = idx; i < zn; i += (long)gridDim.x * 256) zero[i] = 0.f;
  const long row = idx >> 3, total = nrows * a.H;   // nrows = B * Tq, or all stacked rows
  const int d0 = (int)(idx & 7) * 8;
  const bool ok = row < total;
  const int h = ok ? (int)(row % a.H) : 0;
  const long bt = ok ? row / a.H : 0;
  float s = 0.f;
  if (a.dout) {   // (nullptr: the side-stream prologue, which prepares what depends on forward data only)
    float dv[8], ov[8];
    load8<T>((const T*)a.dout + bt * a.ldo + h * DK + d0, dv);
    load8<T>((const T*)a.out + bt * a.ldo + h * DK + d0, ov);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += dv[j] * ov[j];
    s = sum8(s);
  }
  if (!ok) return;
  if (d0 == 0 && a.dout) {
    if (a.nseg > 1) {   // delta of segment s: [utterances, H, T_s] at H * (its first row)
      const SegRef g = seg_of_row(a, bt);
      const long loc = bt - g.row;
      const int i = (int)(loc % g.T), b = (int)(loc / g.T);
      a.delta[g.row * a.H + ((long)b * a.H + h) * g.T + i] = s;
    } else {
      const int i = (int)(bt % a.Tq), b = (int)(bt / a.Tq);
      a.delta[((long)b * a.H + h) * a.Tq + i] = s;
    }
  }
  const long o = (bt * a.H + h) * DK + d0;
  if (qu) {
    float qx[8], u[8], v[8];
    load8<T>((const T*)a.q + bt * a.ldq + h * DK + d0, qx);
#pragma unroll
    for (int j = 0; j < 8; ++j) { u[j] = qx[j] + a.bias_u[h * DK + d0 + j]; v[j] = qx[j] + a.bias_v[h * DK + d0 + j]; }
    store8<T>(qu + o, u);
    store8<T>(qv + o, v);
  }
}

// FW = waves (key tiles) per workgroup: 4 (one workgroup per CU), or 2 with half the LDS (two workgroups per CU) when
// the 4-wave grid would need a second, nearly empty round of workgroups
template <typename T, int FW> struct FusedCfg {
  static constexpr int LD = AttnCfg<T>::LD;
  static constexpr int IMG = DqCfg<T>::IMG_LD;
  // per wave: the dS image [32 keys][IMG] (dK product), the same values once more as the band gradient dG[query][band column
  // c = key - query + 31] ([32][LDG], k-contiguous B operand of the band part of dQ; the entries a lane never writes -- keys
  // outside the tile -- are zeroed once), then a region shared by the 64-row f32 skew tile (before the soft-max) and the P image
  // (after it); the K tile is staged in the same region once, before the sweep.  (dG used to be gathered out of a zero-padded dS
  // image: 32 two-byte LDS reads + 16 packs per lane and step against 16 more two-byte writes + 4 sixteen-byte reads.)
  static constexpr int LDG = 72;
  static constexpr int IMG_DS_BYTES = (32 * IMG + 32 * LDG) * (int)sizeof(T);
  // at the end of a step the region holds the wave's dQ slab, query-major: [32 queries][DQ_LD floats] (64 d + 4 of padding).
  // A lane owns 4 consecutive d of one query after the chained products (accumulator rows r & 3), so the slab is written
  // with 16-byte stores and flushed with 16-byte reads / global stores (the d-major slab took 32 + 32 scalar LDS accesses and
  // 16 four-byte global stores per lane and step: 3.9 k of the step's 11.6 k cycles, in-kernel stamps)
  static constexpr int DQ_LD = 68;
  static constexpr int GS_BYTES = 32 * DQ_LD * 4;  // >= 64 * 32 * 4 (the f32 skew tile before the soft-max)
  static constexpr int WAVE_BYTES = IMG_DS_BYTES + GS_BYTES;
  static constexpr int BAND_ROWS = 32 * FW + 32;
  static constexpr int stage_rows(bool rel) { return rel ? 96 + BAND_ROWS : 64; }
  static constexpr int stage_bytes(bool rel) { return stage_rows(rel) * LD * (int)sizeof(T); }
  static constexpr int smem_bytes(bool rel) { return stage_bytes(rel) + FW * WAVE_BYTES; }
};

#ifdef EMO_ATTN_STAMP
#define EMO_STAMP(k)                                                                                          \
  do {                                                                                                        \
    if (ws.stamp && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)               \
      ws.stamp[step * 13 + (k)] = __builtin_amdgcn_s_memtime();                                               \
  } while (0)
#define EMO_WSTAMP(W, k)                                                                                     \
  do {                                                                                                        \
    if ((W).stamp && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)              \
      (W).stamp[60 * 13 + (k)] = __builtin_amdgcn_s_memtime();                                                \
  } while (0)
#else
#define EMO_STAMP(k) do {} while (0)
#define EMO_WSTAMP(W, k) do {} while (0)
#endif

// the workgroup barrier of the sweep: LDS traffic only (no vmcnt: the prefetch loads, the dS stores and the dQ
// atomics stay in flight across it -- __syncthreads() would drain all three every step)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

template <typename T, bool TR, bool REL, int FW>
__global__ __launch_bounds__(64 * FW, 1) void attn_bwd_fused_kernel(const emoasr_attn_t a_in, const FusedWs ws_in, const int nt) {
  using M_ = Mma<T>;
  using C_ = FusedCfg<T, FW>;
  constexpr int NK = AttnCfg<T>::NK, NS = AttnCfg<T>::NS, LD = C_::LD, IMG = C_::IMG;
  constexpr int VEC = 16 / sizeof(T), PER_ROW = DK / VEC;
  constexpr int NTHR = 64 * FW;
  constexpr int NROWS = C_::stage_rows(REL), NPIECE = NROWS * PER_ROW, PPT = NPIECE / NTHR;  // 16-byte pieces per thread
  static_assert(NPIECE % NTHR == 0, "staging pieces must divide evenly");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, il = lane & 31;
  emoasr_attn_t a = a_in;
  FusedWs ws = ws_in;
  const Blk3 blk = attn_block(nt, a_in.H, a_in.B);
  if (!blk.ok) return;
  int b = blk.z;
  if (a_in.nseg > 1) {   // stacked micro-batches: this workgroup's segment (the scratch images are stacked like the rows)
    const SegRef g = seg_of_slot(a_in, blk.z, b);
    seg_apply<T>(a, g);
    ws.qu = (const T*)ws.qu + g.row * ws.ldqu;
    ws.qv = (const T*)ws.qv + g.row * ws.ldqu;
    ws.dq32 += g.row * (long)(a.H * DK);
    if (ws.dsq) ws.dsq = (T*)ws.dsq + g.row * a.H * ws.ldds;
  }
  const int jblk = blk.x * (32 * FW), h = blk.y;
  if (jblk >= a.Tk) return;   // (a shorter segment of a stacked launch: the grid follows the longest)
  const HeadPtrs hp = head_ptrs<T>(a, b, h);
  const int j0 = jblk + 32 * wave;
  if (jblk >= hp.klen) {  // every key of this block is masked: its dK / dV rows are zero, nothing else to add
    f32x16 z[2];
    zero16(z[0]); zero16(z[1]);
    store_dT<T>((T*)hp.dk, a.ldk, j0, a.Tk, z, 1.f, lane);
    store_dT<T>((T*)hp.dv, a.ldv, j0, a.Tk, z, 1.f, lane);
    return;
  }
  const bool live = j0 < hp.klen;  // a dead wave still stages, joins the barriers and flushes

  T* stage0 = reinterpret_cast<T*>(smem);
  char* mine = smem + C_::stage_bytes(REL) + wave * C_::WAVE_BYTES;
  T* img_ds = reinterpret_cast<T*>(mine);            // [32 keys][IMG]
  T* img_g = img_ds + 32 * IMG;                      // [32 queries][LDG]
  float* Gs = reinterpret_cast<float*>(mine + C_::IMG_DS_BYTES);  // [64 band rows][32 queries] f32 ...
  T* img_p = reinterpret_cast<T*>(mine + C_::IMG_DS_BYTES);       // ... later the P image [32][IMG]
  for (int i = lane; i < C_::IMG_DS_BYTES / 4; i += 64) reinterpret_cast<unsigned*>(img_ds)[i] = 0u;
  // the four slab bases, for the flush (slab w = wave w's skew region)
  const float* slab0 = reinterpret_cast<const float*>(smem + C_::stage_bytes(REL) + C_::IMG_DS_BYTES);
  constexpr int SLAB_STRIDE = C_::WAVE_BYTES / 4;

  // ---- per-(b,h) operand bases ----------------------------------------------------------------------------
  const long ho = (long)h * DK;
  const T* qu_base = (const T*)ws.qu + (long)b * a.Tq * ws.ldqu + ho;
  const T* qv_base = REL ? (const T*)ws.qv + (long)b * a.Tq * ws.ldqu + ho : qu_base;
  const __amdgpu_buffer_rsrc_t rsQu = make_rsrc(qu_base), rsQv = make_rsrc(qv_base), rsDo = make_rsrc(hp.dout),
                               rsP = make_rsrc(REL ? hp.pos : hp.dout);
  const int nstep = (a.Tq + 31) / 32;

  // ---- one-step-ahead operand fetch (whole block) -----------------------------------------------------------
  // stage rows: [0,32) Q+u, then (REL) [32,64) Q+v, [64,96) dO, [96,96+BAND_ROWS) band; (!REL) [32,64) dO
  constexpr int RPP = NTHR / PER_ROW;  // stage rows covered by one piece index (32 or 16)
  static_assert(32 % RPP == 0, "a piece index must stay inside one 32-row operand tile");
  constexpr int NMAT = REL ? 3 : 2;  // 32-row tiles ahead of the band: Q+u, (Q+v,) dO
  Vec16<T> pre[PPT];
  float pre_lse, pre_del, nxt_lse = INFINITY, nxt_del = 0.f;
  const __amdgpu_buffer_rsrc_t rsL = make_rsrc(hp.lse), rsD = make_rsrc(hp.delta);
  auto fetch = [&](const int step) {
    const int i0 = step * 32;
    const bool on = step < nstep;
    const int trow = tid / PER_ROW, piece = (tid % PER_ROW) * VEC;
    {
      const unsigned o = on && i0 + il < a.Tq ? (unsigned)((i0 + il) * 4) : EMO_OOB;
      pre_lse = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsL, o, 0, 0));
      pre_del = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsD, o, 0, 0));
    }
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      const int srow = p * RPP;  // first stage row of this piece index (compile-time after unrolling)
      if (srow < 32 * NMAT) {
        const int mat = srow / 32, i = i0 + srow % 32 + trow;
        const bool ok = on && i < a.Tq;
        if (mat == 0) pre[p] = buf_load16<T>(rsQu, ok ? (unsigned)(((long)i * ws.ldqu + piece) * sizeof(T)) : EMO_OOB);
        else if (REL && mat == 1) pre[p] = buf_load16<T>(rsQv, ok ? (unsigned)(((long)i * ws.ldqu + piece) * sizeof(T)) : EMO_OOB);
        else pre[p] = buf_load16<T>(rsDo, ok ? (unsigned)(((long)i * a.ldo + piece) * sizeof(T)) : EMO_OOB);
      } else {
        const int r = a.Tq - 32 - i0 + jblk + (srow - 32 * NMAT) + trow;
        const bool ok = on && r >= 0 && r < 2 * a.Tq - 1;
        pre[p] = buf_load16<T>(rsP, ok ? (unsigned)(((long)r * a.ldp + piece) * sizeof(T)) : EMO_OOB);
      }
    }
  };
  auto stash = [&]() {
    T* st = stage0;
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      const int pid = tid + NTHR * p;
      store16(st + (pid / PER_ROW) * LD + (pid % PER_ROW) * VEC, pre[p]);
    }
    nxt_lse = pre_lse;
    nxt_del = pre_del;
  };

  // ---- stationary operands of this wave's key tile ---------------------------------------------------------
  const int kj_lane = j0 + il;
  typename M_::Frag kfA[NK], vfA[NK], kT[2][NS];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    kfA[kk] = frag_global<T>((const T*)hp.k, a.ldk, kj_lane, live && kj_lane < a.Tk, kk, lane, nullptr);
    vfA[kk] = frag_global<T>((const T*)hp.v, a.ldv, kj_lane, live && kj_lane < a.Tk, kk, lane, nullptr);
  }
  {
    T* Ks = reinterpret_cast<T*>(Gs);  // [32][LD] in the wave's skew region
    stage_rows<T, 32>(Ks, (const T*)hp.k, a.ldk, j0, 0, live ? a.Tk : 0, lane, nullptr);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int ks = 0; ks < NS; ++ks) kT[dt][ks] = chain_a<T, TR>(Ks, ks, 32 * dt, lane);
    __builtin_amdgcn_wave_barrier();
  }
  f32x16 dk[2], dv[2], csum;
  zero16(dk[0]); zero16(dk[1]); zero16(dv[0]); zero16(dv[1]); zero16(csum);
  if (!live)  // a dead wave's slab stays zero for the whole sweep
    for (int i = lane; i < 32 * C_::DQ_LD; i += 64) Gs[i] = 0.f;
  // sum of the waves' dQ^T slabs of the step that just ended -> this key block's f32 partial (rows of 64 d = 256 B, plain
  // stores: every (key block, query row) has exactly one writer, so dQ is bit-reproducible); wave w takes rows w, w + FW, ...
  // flush lane map: 4 query rows per pass (lane >> 4), 4 consecutive d per lane (lane & 15)
  const int f_row = lane >> 4, f_d = 4 * (lane & 15);
  float* dq_part = ws.dq32 + (long)blk.x * ws.dq_slab + (long)b * a.Tq * (a.H * DK) + ho + f_d;
  f32x4 dq_colsum4 = f32x4{0.f, 0.f, 0.f, 0.f};  // sum over this lane's query rows of dQ[:, f_d .. f_d + 3]
  auto flush = [&](const int ib) {
#pragma unroll
    for (int q0 = 4 * wave; q0 < 32; q0 += 4 * FW) {
      const int q = q0 + f_row;
      f32x4 v = *reinterpret_cast<const f32x4*>(slab0 + q * C_::DQ_LD + f_d);
#pragma unroll
      for (int w = 1; w < FW; ++w) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(slab0 + w * SLAB_STRIDE + q * C_::DQ_LD + f_d);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += o[e];
      }
      if (ib + q < a.Tq) {
        *reinterpret_cast<f32x4*>(dq_part + (long)(ib + q) * (a.H * DK)) = v;
#pragma unroll
        for (int e = 0; e < 4; ++e) dq_colsum4[e] += v[e];
      }
    }
  };
  fetch(0);
  stash();
  fetch(1);
  __syncthreads();
  const bool full_tile = j0 + 32 <= hp.klen;  // no masked key in this wave's tile
  const float c_exp = a.scale * 1.4426950408889634f;

  for (int step = 0; step < nstep; ++step) {
    const int i0 = step * 32;
    const T* st = stage0;
    const T* Qus = st;
    const T* Qvs = st + 32 * LD;
    const T* dOs = st + (REL ? 64 : 32) * LD;
    const T* Bs = st + (96 + 32 * wave) * LD;  // this wave's 64 band rows (REL)
    EMO_STAMP(0);
    if (live) {
      const int qi = i0 + il;
      const bool qval = qi < a.Tq;
      // p = exp2(c_exp * s - lse * log2 e); rows without any valid key (lse = -inf) and padding queries get +inf -> p = 0
      const float lse2 = (qval && nxt_lse != -INFINITY) ? nxt_lse * 1.4426950408889634f : INFINITY;
      const float del_q = nxt_del;
      const uint64_t drop_base = drop_index(a, b, h, qi, 0);
      // S^T = K (Q+u)^T, band G^T = band (Q+v)^T, dP^T = V dO^T.  One wave per SIMD: nothing hides an LDS round trip or a
      // dependent MFMA, so every operand fragment of the phase is requested first and the four accumulation chains are
      // interleaved (each MFMA waited for the read issued right before it and for its predecessor in the chain otherwise)
      typename M_::Frag fq[NK], fo[NK], fv[NK], fb0[NK], fb1[NK];
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) fq[kk] = M_::load_kc(Qus, LD, 0, kk * M_::KSTEP, lane);
      if constexpr (REL) {
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
          fv[kk] = M_::load_kc(Qvs, LD, 0, kk * M_::KSTEP, lane);
          fb0[kk] = M_::load_kc(Bs, LD, 0, kk * M_::KSTEP, lane);
          fb1[kk] = M_::load_kc(Bs + 32 * LD, LD, 0, kk * M_::KSTEP, lane);
        }
      }
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) fo[kk] = M_::load_kc(dOs, LD, 0, kk * M_::KSTEP, lane);
      __builtin_amdgcn_sched_barrier(0);
      f32x16 s, dp;
      zero16(s); zero16(dp);
      if constexpr (REL) {
        f32x16 g0, g1;
        zero16(g0); zero16(g1);
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
          s = M_::mma(kfA[kk], fq[kk], s);
          g0 = M_::mma(fb0[kk], fv[kk], g0);  // g[c][i]
          g1 = M_::mma(fb1[kk], fv[kk], g1);
          dp = M_::mma(vfA[kk], fo[kk], dp);
        }
        // (the optimiser sinks the S chain below the skew writes, into the shadow of their LDS round trip: pinning it up here with
        // an empty asm was measured 85 cycles per step slower)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          Gs[c_row(r, lane) * 32 + il] = g0[r];
          Gs[(32 + c_row(r, lane)) * 32 + il] = g1[r];
        }
        __builtin_amdgcn_wave_barrier();
        // element (key jl, query il) sits in band column c = 31 - il + jl
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] += Gs[(31 - il + c_row(r, lane)) * 32 + il];
        __builtin_amdgcn_wave_barrier();
      } else {
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
          s = M_::mma(kfA[kk], fq[kk], s);
          dp = M_::mma(vfA[kk], fo[kk], dp);
        }
      }
      EMO_STAMP(1);
      EMO_STAMP(2);
      // (placing these ~300 VALU instructions ahead of the MFMA chains above moved their 1.5 k cycles, it did not hide them)
      f32x16 ds, dsc;
      if (a.drop_p > 0.f) {
        const float keep = 1.f / (1.f - a.drop_p);
        const uint32_t thr = dropout_thr(a.drop_p);
        // accumulator rows 4 g .. 4 g + 3 are keys j0 + 8 g + 4 (lane >> 5) + 0 .. 3: two hash pairs (drop_base, j0 even)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint64_t pr = (drop_base + (uint64_t)(j0 + 8 * g + 4 * (lane >> 5))) >> 1;
          bool k0, k1, k2, k3;
          dropout_keep2(a.seed, pr, thr, k0, k1);
          dropout_keep2(a.seed, pr + 1, thr, k2, k3);
          dsc[4 * g] = k0 ? keep : 0.f; dsc[4 * g + 1] = k1 ? keep : 0.f;
          dsc[4 * g + 2] = k2 ? keep : 0.f; dsc[4 * g + 3] = k3 ? keep : 0.f;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) dsc[r] = 1.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = __builtin_amdgcn_exp2f(s[r] * c_exp - lse2);
        if (!full_tile) p = (j0 + c_row(r, lane) >= hp.klen) ? 0.f : p;
        ds[r] = p * (dp[r] * dsc[r] - del_q) * a.scale;
        csum[r] += ds[r];
        img_ds[c_row(r, lane) * IMG + il] = from_f32<T>(ds[r]);
        if constexpr (REL) img_g[il * (C_::LDG - 1) + c_row(r, lane) + 31] = from_f32<T>(ds[r]);  // [il][key - il + 31]
        img_p[c_row(r, lane) * IMG + il] = from_f32<T>(p * dsc[r]);
      }
      __builtin_amdgcn_wave_barrier();
      EMO_STAMP(3);
      if constexpr (REL) {
        // dS for the dpos pass: query-major rows, 4 consecutive keys (8 bytes) per store
        if (qval) {
          T* drow = (T*)ws.dsq + (((long)b * a.H + h) * a.Tq + qi) * ws.ldds + j0 + 4 * (lane >> 5);
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            bf16x4 v4;
#pragma unroll
            for (int e = 0; e < 4; ++e) v4[e] = (bf16)ds[4 * g4 + e];
            *reinterpret_cast<bf16x4*>(drow + 8 * g4) = v4;
          }
        }
      }
      EMO_STAMP(4);
      // dQ^T partial = K^T dS^T (+ band^T unskew(dS^T));  dV^T += dO^T P,  dK^T += (Q+u)^T dS (sum over the query index:
      // operands from the bf16 images).  Reads first, as above: the band rows and the un-skewed dS ahead of the four register-
      // operand MFMAs, the dV / dK operands ahead of the band MFMAs.
      typename M_::Frag dgf[2][NS], fba[2][2][NS];
      if constexpr (REL) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int ks = 0; ks < NS; ++ks) {
            dgf[ct][ks] = M_::load_kc(img_g, C_::LDG, 0, 32 * ct + ks * M_::KSTEP, lane);  // dG^T[c][i], c contiguous
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
              fba[ct][dt][ks] = M_::template load_km<TR>(Bs + 32 * ct * LD, LD, ks * M_::KSTEP, 32 * dt, lane);
          }
      }
      typename M_::Frag dsf[NS];
#pragma unroll
      for (int ks = 0; ks < NS; ++ks) dsf[ks] = chain_b<T>(ds, ks);
      __builtin_amdgcn_sched_barrier(0);
      f32x16 dq[2];
      zero16(dq[0]); zero16(dq[1]);
#pragma unroll
      for (int ks = 0; ks < NS; ++ks)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) dq[dt] = M_::mma(kT[dt][ks], dsf[ks], dq[dt]);
      typename M_::Frag fdo[2][NS], fqu[2][NS], fp[NS], fd[NS];
#pragma unroll
      for (int ks = 0; ks < NS; ++ks) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          fdo[dt][ks] = M_::template load_km<TR>(dOs, LD, ks * M_::KSTEP, 32 * dt, lane);
          fqu[dt][ks] = M_::template load_km<TR>(Qus, LD, ks * M_::KSTEP, 32 * dt, lane);
        }
        fp[ks] = M_::load_kc(img_p, IMG, 0, ks * M_::KSTEP, lane);
        fd[ks] = M_::load_kc(img_ds, IMG, 0, ks * M_::KSTEP, lane);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (REL) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int ks = 0; ks < NS; ++ks)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) dq[dt] = M_::mma(fba[ct][dt][ks], dgf[ct][ks], dq[dt]);
      }
      EMO_STAMP(5);
#pragma unroll
      for (int ks = 0; ks < NS; ++ks)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dv[dt] = M_::mma(fdo[dt][ks], fp[ks], dv[dt]);
          dk[dt] = M_::mma(fqu[dt][ks], fd[ks], dk[dt]);
        }
      __builtin_amdgcn_wave_barrier();
      EMO_STAMP(6);
      // the P image (start of the skew region) has been consumed: the region now takes this wave's dQ^T slab
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)   // accumulator rows 4 g4 .. 4 g4 + 3 = d 32 dt + 8 g4 + 4 (lane >> 5) + 0 .. 3 of query il
          *reinterpret_cast<f32x4*>(Gs + il * C_::DQ_LD + 32 * dt + 8 * g4 + 4 * (lane >> 5)) =
              f32x4{dq[dt][4 * g4], dq[dt][4 * g4 + 1], dq[dt][4 * g4 + 2], dq[dt][4 * g4 + 3]};
    }
    EMO_STAMP(7);
    lds_barrier();  // every wave has read the stage and written its slab
    EMO_STAMP(8);
    flush(i0);
    EMO_STAMP(9);
    stash();         // tiles of step+1 (fetched during the previous step)
    EMO_STAMP(10);
    fetch(step + 2);
    EMO_STAMP(11);
    lds_barrier();  // stage ready; every slab has been read: the skew regions may be written again
    EMO_STAMP(12);
  }
  store_dT<T>((T*)hp.dk, a.ldk, j0, a.Tk, dk, 1.f, lane);  // (a dead wave stores zeros)
  store_dT<T>((T*)hp.dv, a.ldv, j0, a.Tk, dv, 1.f, lane);
  if (REL && a.dbias_u) {
    // dbias_u[d] += sum_j colsum_j K[j][d]  (colsum_j = sum_i dS[i][j]);  dbias_v[d] += colsum(dQ)[d] - that
    float acc = 0.f;
    if (live) {
      float* cs = Gs;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = csum[r];
        v = half_sum32(v);
        if (il == 31) cs[c_row(r, lane)] = v;
      }
      __builtin_amdgcn_wave_barrier();
      const __amdgpu_buffer_rsrc_t rsK = make_rsrc(hp.k);
#pragma unroll 8
      for (int j = 0; j < 32; ++j) {
        const int kj = j0 + j;
        acc += cs[j] * buf_load_f32<T>(rsK, kj < a.Tk ? (unsigned)(((long)kj * a.ldk + lane) * sizeof(T)) : EMO_OOB);
      }
      atomicAdd(&a.dbias_u[h * DK + lane], acc);
    }
    // colsum(dQ)[d = lane]: fold the four row groups of the flush map, then pick component lane & 3 of lane (lane >> 2)'s sums
    float dq_colsum = 0.f;
    {
      f32x4 t = dq_colsum4;
#pragma unroll
      for (int e = 0; e < 4; ++e) { t[e] += __shfl_xor(t[e], 16, 64); t[e] += __shfl_xor(t[e], 32, 64); }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ve = __shfl(t[e], lane >> 2, 64);
        if ((lane & 3) == e) dq_colsum = ve;
      }
    }
    atomicAdd(&a.dbias_v[h * DK + lane], dq_colsum - acc);
  }
}

// dpos[r, h*64+d] += sum_{b,i} dS[b,h,i, j = i-(Tq-1)+r] (Q+v)[b,i,h*64+d]: block = (32 table rows, head, batch chunk);
// the waves take the (batch, query tile) items whose diagonal band meets valid keys round-robin.  Per item: the band of
// dS is gathered with 2-byte loads (consecutive lanes = consecutive keys) as the A operand, the 32x64 (Q+v) tile comes in
// with 16-byte loads and goes through wave-private LDS (transposed reads) as the B operand; the next item's loads are in
// flight while this one is multiplied.  Waves are reduced through LDS before one set of atomics (rows of 64 d).
// ------------------------------------------------------------------------------------------------------------------------------------
// Position-table gradient, round 6 (bf16): dpos[r] = sum_{b,i} dS_b[i, j = i - (T-1) + r] (Q+v)_b[i] from the stored dS image, one
// DIAGONAL of 32 x 32 tiles per workgroup.  Every tile (i0, j0 = i0 + 32 dg) of a diagonal maps to the SAME 64 table rows
// rbase + c, rbase = T - 32 + 32 dg, c = 31 - il + jl, so a wave keeps ONE [64 c][64 d] accumulator for all the tiles and all the
// utterances it visits and the workgroup leaves with one 16 KB atomic flush.  Per tile: the dS tile (2 KB) and the (Q+v) tile
// (4 KB) arrive by 16-byte loads one item ahead; the dS tile sits in wave-private LDS between two zero-filled 32-column wings, so
// the skewed operand dG^T[c][i] = dS[i][c + i - 31] is 32 two-byte LDS reads at COMPILE-TIME offsets from one per-lane base (the
// wings supply the zeros of the band's corners); (Q+v) is read k-major with transposing reads.  the round 2-5 kernel (attn_bwd_dpos2_kernel) gathered
// the diagonal elements from global memory with two-byte loads: 84 non-matrix instructions per MFMA, 141 us per layer launch at
// the bench's shapes (profiles/r05_attn_counters.txt).
// ------------------------------------------------------------------------------------------------------------------------------------
struct Dpos3Plan { int nseg, nch; int wg0[EMOASR_MAX_SEGMENTS + 1]; };   // workgroups of segment s: [wg0[s], wg0[s+1]) = H x (2 nt - 1) x nch
template <typename T, bool TR>
__global__ __launch_bounds__(256) void attn_bwd_dpos3_kernel(const emoasr_attn_t a_in, const FusedWs ws_in, const Dpos3Plan plan) {
  using M_ = Mma<T>;
  static_assert(sizeof(T) == 2, "bf16 only");
  constexpr int LD = AttnCfg<T>::LD;            // (Q+v) tile row stride
  constexpr int LDS_ = 104;                     // dS tile row stride: [32 zeros | 32 keys | 32 zeros | pad]
  constexpr int DS_BYTES = 32 * LDS_ * 2, QV_BYTES = 32 * LD * 2, WAVE_BYTES = DS_BYTES + QV_BYTES;
  constexpr int RED_BYTES = 4 * 2 * 16 * 64 * 4;   // one 32-row table tile of all four waves at a time
  __shared__ __attribute__((aligned(16))) char smem[(4 * WAVE_BYTES > RED_BYTES) ? 4 * WAVE_BYTES : RED_BYTES];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), il = lane & 31, hh = lane >> 5;
  emoasr_attn_t a = a_in;
  FusedWs ws = ws_in;
  int sgi = 0;
  for (int k = 1; k < EMOASR_MAX_SEGMENTS; ++k) sgi += (k < plan.nseg && (int)blockIdx.x >= plan.wg0[k]) ? 1 : 0;
  int w = blockIdx.x - plan.wg0[sgi];
  if (a_in.nseg > 1) {
    const SegRef g = seg_ref(a_in, sgi);
    seg_apply<T>(a, g);
    ws.qv = (const T*)ws.qv + g.row * ws.ldqu;
    ws.dsq = (T*)ws.dsq + g.row * a.H * ws.ldds;
  }
  const int nt = (a.Tq + 31) / 32, ndiag = 2 * nt - 1;
  const int chunk = w % plan.nch; w /= plan.nch;
  const int dgi = w % ndiag, h = w / ndiag;
  const int dg = dgi - (nt - 1);                       // j0 = i0 + 32 dg
  const int len = nt - (dg < 0 ? -dg : dg), t_lo = dg < 0 ? -dg : 0;   // tiles of the diagonal: i0 = 32 (t_lo + k), k < len
  const int bper = (a.B + plan.nch - 1) / plan.nch, b_lo = chunk * bper, b_hi = min(a.B, b_lo + bper);
  const int nitem = (b_hi - b_lo) * len;
  T* dss = reinterpret_cast<T*>(smem + wave * WAVE_BYTES);
  T* qs = reinterpret_cast<T*>(smem + wave * WAVE_BYTES + DS_BYTES);
  // the wings: written once
  for (int v = lane; v < 32 * 9; v += 64) {   // per row: columns [0,32) and [64,104) = 4 + 5 pieces of 8
    const int row = v / 9, pc = v % 9;
    Vec16<T> z; z.zero();
    store16(dss + row * LDS_ + (pc < 4 ? 8 * pc : 64 + 8 * (pc - 4)), z);
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) { zero16(acc[t][0]); zero16(acc[t][1]); }
  const long ldqu2 = ws.ldqu * 2, ldds2 = ws.ldds * 2;
  Vec16<T> xs[2], xq[4];
  auto fetch = [&](const int item) {
    const bool on = item < nitem;
    const int bo = on ? item / len : 0, k = item - bo * len, b = b_lo + bo;
    const int i0 = 32 * (t_lo + k), j0 = i0 + 32 * dg;
    const int klen = a.klens ? min(a.klens[b], a.Tk) : a.Tk;
    const bool live = on && j0 < klen;   // key tiles without a valid key were never written
    const __amdgpu_buffer_rsrc_t rsS = make_rsrc((const T*)ws.dsq + (((long)b * a.H + h) * a.Tq + i0) * ws.ldds + j0);
    const __amdgpu_buffer_rsrc_t rsQ = make_rsrc((const T*)ws.qv + ((long)b * a.Tq + i0) * ws.ldqu + (long)h * DK);
#pragma unroll
    for (int q = 0; q < 2; ++q) {   // dS tile: 32 rows x 4 pieces
      const int p = lane + 64 * q, row = p >> 2, pc = p & 3;
      xs[q] = buf_load16<T>(rsS, (live && i0 + row < a.Tq) ? (unsigned)(row * ldds2 + 16 * pc) : EMO_OOB);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // (Q+v) tile: 32 rows x 8 pieces
      const int p = lane + 64 * q, row = p >> 3, pc = p & 7;
      xq[q] = buf_load16<T>(rsQ, (live && i0 + row < a.Tq) ? (unsigned)(row * ldqu2 + 16 * pc) : EMO_OOB);
    }
  };
  // per-lane base of the skewed reads: element e of k-step ks of table tile t is dS[i = 16 ks + 8 hh + e][jl = 32 t + il + i - 31],
  // stored at column 32 + jl
  const T* gbase = dss + (8 * hh) * (LDS_ + 1) + il + 1;
  fetch(wave);
  for (int item = wave; item < nitem; item += 4) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int p = lane + 64 * q;
      store16(dss + (p >> 2) * LDS_ + 32 + 8 * (p & 3), xs[q]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p = lane + 64 * q;
      store16(qs + (p >> 3) * LD + 8 * (p & 7), xq[q]);
    }
    fetch(item + 4);
    __builtin_amdgcn_wave_barrier();
    typename M_::Frag fa[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) fa[t][ks][e] = gbase[(16 * ks + e) * (LDS_ + 1) + 32 * t];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const typename M_::Frag fb = M_::template load_km<TR>(qs, LD, 16 * ks, 32 * dt, lane);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t][dt] = M_::mma(fa[t][ks], fb, acc[t][dt]);
      }
    __builtin_amdgcn_wave_barrier();
  }
  const int rbase = a.Tq - 32 + 32 * dg;
  float* red = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    __syncthreads();
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int rg = 0; rg < 16; ++rg) red[((wave * 2 + dt) * 16 + rg) * 64 + lane] = acc[t][dt][rg];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {   // 2048 values of the tile, 8 per thread
      const int idx = threadIdx.x + 256 * k, ln = idx & 63, rg = (idx >> 6) & 15, dt = idx >> 10;
      float v = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) v += red[((wv * 2 + dt) * 16 + rg) * 64 + ln];
      const int row = rbase + 32 * t + c_row(rg, ln);
      if (row >= 0 && row < 2 * a.Tq - 1 && nitem > 0) atomicAdd(&a.dpos[(long)row * (a.H * DK) + h * DK + 32 * dt + (ln & 31)], v);
    }
  }
}

// ====================================================================================
// backward, two passes (round 4; bf16) -- the default.  The single-pass kernel above has ONE stationary side (the keys), so the
// other side's gradient (dQ) leaves it as per-key-block f32 partial slabs that a finalize launch sums, P and dS cross LDS as
// transposed images, and its 98 KB of LDS / 232 registers allow one workgroup per CU: one wave per SIMD, nothing hides anything.
// Here every score tile is recomputed TWICE (the matrix pipe sat at 6 %: the flops are free), each time in the orientation
// whose accumulator is directly the next product's operand:
//   attn_bwd_kv_kernel  key-stationary, S[query][key] with the KEY on the lane: P and dS chain straight into dV^T += dO^T P and
//                       dK^T += (Q+u)^T dS (sums over the query index, which sits in the accumulator registers); per-key column
//                       sums of dS are one add per element (dbias_u).  No image, no dQ.
//   attn_bwd_q_kernel   query-stationary, S^T[key][query] with the QUERY on the lane: dS^T chains into dQ^T += K^T dS^T, the band
//                       part through the un-skewed dG image (as before); dQ is complete in ONE workgroup and is stored once, in the
//                       compute dtype: no partial slabs, no finalize pass, bit-reproducible.  Writes the dS image the position-table
//                       gradient (attn_bwd_dpos3_kernel) walks.
// Both: 4 waves per workgroup, the per-step operand tiles staged once per workgroup (registers one step ahead -> LDS), 65-70 KB
// of LDS and <= 256 registers: TWO workgroups per CU, two waves per SIMD -- one wave's soft-max / dropout VALU work runs under
// the other's MFMAs and LDS round trips.
// Why the dS image stays (review item "no dS image"): dpos[r] = sum_{b,i} dS_b[i, i-(T-1)+r] (Q+v)_b[i] sums over the utterances
// of a segment; a workgroup that holds a (b, h) tile can only add its [T+32 .. 2T][64] f32 partial with atomics -- 110-440 MB of
// float atomics per layer launch at the chip's 1.3 TB/s atomic rate, against 110 MB written + read at stream rate for the bf16
// image -- or a third recomputation sweep that is table-row-stationary (only half of each recomputed tile lies on its band).
// ====================================================================================
template <typename T, int FW> struct SplitCfg {
  static constexpr int LD = AttnCfg<T>::LD;
  static constexpr int BAND_ROWS = 32 * FW + 32;
  static constexpr int LDG = 72;                     // dG image row stride ([32 queries][64 band columns] + padding)
  static constexpr int LDGS = 68;                    // f32 skew tile G[query][band column] row stride (Q pass; see attn_fwd_kernel)
  static constexpr int GS_BYTES = 32 * LDGS * 4;     // per wave: the f32 skew tile (the Q pass's dG image reuses it)
  static constexpr int ROWC_BYTES = 64 * 4 + FW * 32 * 4;   // KV pass: lse * log2(e), delta and the waves' keep-mask words of the step's 32 queries
  static constexpr int RING_BLOCKS = FW + 2;         // KV pass: the band as a ring of 32-row blocks (FW + 1 in use, one being refilled)
  static constexpr int kv_rows(bool rel) { return rel ? 96 + 32 * RING_BLOCKS : 64; }   // Q+u, Q+v, dO, band ring | Q, dO
  static constexpr int q_rows(bool rel) { return rel ? 64 + 32 * RING_BLOCKS : 64; }    // K, V, band ring | K, V
  static constexpr int kv_stage_bytes(bool rel) { return kv_rows(rel) * LD * (int)sizeof(T) + ROWC_BYTES; }
  static constexpr int q_stage_bytes(bool rel) { return q_rows(rel) * LD * (int)sizeof(T); }
  static constexpr int wave_bytes(bool rel) { return rel ? GS_BYTES : 256; }
  static constexpr int kv_smem(bool rel) { return kv_stage_bytes(rel) + 2 * 32 * FW * LD * (int)sizeof(T); }   // + the workgroup's K / V tiles
  static constexpr int q_smem(bool rel) { return q_stage_bytes(rel) + FW * wave_bytes(rel); }
};

// The attention-dropout keep mask of one launch as BITS.  The forward regenerates the counter-based mask inline (one hash per pair
// of keys, common.h: dropout_keep2); so did the single-pass backward -- and in the two recomputation passes of the two-pass backward
// the hash was the largest VALU item of a step (PMC: 566 / 424 VALU instructions per 32 x 32 tile in the key- / query-stationary
// pass, ~420 / ~220 of them the mask: 26 per element where the keys of a pair sit on neighbouring lanes, 13 where they sit in
// neighbouring registers).  Here every pair is hashed ONCE per layer and step, by a kernel that does nothing else (one thread per
// 32-key word, 16 pair hashes), and both passes test a bit: 2 instructions per element.  Same mask, bit for bit.
template <typename T>
__global__ __launch_bounds__(1024) void attn_dropmask_kernel(const emoasr_attn_t a_in, unsigned* __restrict__ mask, const int nw,
                                                            const long nrows) {
  // block = RPB consecutive rows x H heads x nw words: (threadIdx.x, threadIdx.y, threadIdx.z) = (word, head, row in block), so
  // that no thread divides (the first version decoded a linear index with five integer divisions per thread: 41 us per launch)
  const long row = (long)blockIdx.x * blockDim.z + threadIdx.z;
  const int w = threadIdx.x, h = threadIdx.y;
  if (row >= nrows || w >= nw) return;
  emoasr_attn_t a = a_in;
  int b, i;
  if (a_in.nseg > 1) {
    const SegRef g = seg_of_row(a_in, row);
    seg_apply<T>(a, g);
    const long loc = row - g.row;
    b = (int)(loc / g.T); i = (int)(loc - (long)b * g.T);
  } else {
    b = (int)(row / a.Tq); i = (int)(row - (long)b * a.Tq);
  }
  const int klen = a.klens ? min(a.klens[b], a.Tk) : a.Tk;
  if (32 * w >= klen) return;   // no valid key in this word: never read
  const uint32_t thr = dropout_thr(a.drop_p);
  const uint64_t pr0 = drop_index(a, b, h, i, 32 * w) >> 1;
  unsigned bits = 0u;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    bool k0, k1;
    dropout_keep2(a.seed, pr0 + (uint64_t)k, thr, k0, k1);
    bits |= (k0 ? 1u : 0u) << (2 * k);
    bits |= (k1 ? 1u : 0u) << (2 * k + 1);
  }
  mask[(row * a_in.H + h) * nw + w] = bits;
}

#ifndef EMO_KV_DBG
#define EMO_KV_DBG 0   // timing ablations of attn_bwd_kv_kernel, build variants only (results are then wrong): 1 no band loads, 2 no
#endif                 // lane rotation, 4 no query-side loads.  (As a kernel ARGUMENT -- rounds 4-5 -- every use was a scalar branch:
                       // each of the 32 rotations of a step sat in a basic block of its own.)
template <typename T, bool TR, bool REL, int FW>
__global__ __launch_bounds__(64 * FW, 2) void attn_bwd_kv_kernel(const emoasr_attn_t a_in, const FusedWs ws_in, const int nt) {
  constexpr int dbg = EMO_KV_DBG;
  using M_ = Mma<T>;
  using C_ = SplitCfg<T, FW>;
  constexpr int NK = AttnCfg<T>::NK, NS = AttnCfg<T>::NS, LD = C_::LD;
  constexpr int VEC = 16 / sizeof(T), PER_ROW = DK / VEC;
  constexpr int NTHR = 64 * FW;
  constexpr int NROWS = C_::kv_rows(REL);
  // pieces a thread stages per step: the 32-row tiles Q+u, (Q+v,) dO and -- relative positions -- ONE new 32-row block of the band
  // (round 6: the union of the waves' bands moves down by 32 rows per step, so 128 of its 160 rows are already in LDS; the band is a
  // ring of FW + 2 blocks, the new block lands in the slot no wave reads this step.  20 KB of the 32 KB a step fetched were
  // re-fetched band rows, and the fetch + stash phases were two thirds of a step: profiles/r06_attn_phases.txt)
  constexpr int PPT = (REL ? 3 : 2) + (REL ? 1 : 0), NRING = C_::RING_BLOCKS;
  static_assert(NTHR / PER_ROW == 32, "one piece per thread covers one 32-row tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, il = lane & 31, hh = lane >> 5;
  emoasr_attn_t a = a_in;
  FusedWs ws = ws_in;
  EMO_WSTAMP(ws_in, 0);
  const Blk3 blk = attn_block(nt, a_in.H, a_in.B);
  if (!blk.ok) return;
  int b = blk.z;
  if (a_in.nseg > 1) {
    const SegRef g = seg_of_slot(a_in, blk.z, b);
    seg_apply<T>(a, g);
    ws.qu = (const T*)ws.qu + g.row * ws.ldqu;
    ws.qv = (const T*)ws.qv + g.row * ws.ldqu;
    if (ws.mask) ws.mask += g.row * a.H * ws.mask_nw;
    if (ws.dsq) ws.dsq = (T*)ws.dsq + g.row * a.H * ws.ldds;   // the dS image this pass writes (R6), as the query pass reads it
  }
  const int jblk = blk.x * (32 * FW), h = blk.y;
  if (jblk >= a.Tk) return;
  const HeadPtrs hp = head_ptrs<T>(a, b, h);
  const int j0 = jblk + 32 * wave;
  if (jblk >= hp.klen) {  // every key of this block is masked: its dK / dV rows are zero
    f32x16 z[2];
    zero16(z[0]); zero16(z[1]);
    store_dT<T>((T*)hp.dk, a.ldk, j0, a.Tk, z, 1.f, lane);
    store_dT<T>((T*)hp.dv, a.ldv, j0, a.Tk, z, 1.f, lane);
    return;
  }
  const bool live = j0 < hp.klen;  // a dead wave still stages and joins the barriers

  // LDS: the step's query-side tiles + band (one buffer, refilled between two barriers), the row constants and keep-mask words of
  // the step, and the workgroup's own K / V tiles (staged once: as registers they pushed the kernel past 256 and into scratch)
  T* stage0 = reinterpret_cast<T*>(smem);
  float* rowc = reinterpret_cast<float*>(smem + NROWS * LD * (int)sizeof(T));   // [0,32) lse * log2 e (+inf: no contribution), [32,64) delta
  unsigned* maskw = reinterpret_cast<unsigned*>(rowc + 64);   // [FW waves][32 queries]: the keep-mask words of the step's rows for each wave's key tile
  T* Kt = reinterpret_cast<T*>(smem + C_::kv_stage_bytes(REL)) + 32 * wave * LD;   // this wave's K tile [32][LD]
  T* Vt = Kt + 32 * FW * LD;

  const long ho = (long)h * DK;
  const T* qu_base = (const T*)ws.qu + (long)b * a.Tq * ws.ldqu + ho;
  const T* qv_base = REL ? (const T*)ws.qv + (long)b * a.Tq * ws.ldqu + ho : qu_base;
  // descriptors sized to the valid rows (queries < Tq, table rows < 2 Tq - 1): a row guard is the descriptor's bounds check, a
  // fetch is PPT loads at lane-constant offsets + a multiple of the row stride (see attn_fwd_kernel)
  const unsigned qstride = (unsigned)ws.ldqu * 2u, ostride = (unsigned)a.ldo * 2u, pstride = (unsigned)a.ldp * 2u;
  const unsigned qbytes = (unsigned)(a.Tq - 1) * qstride + DK * 2u;
  const __amdgpu_buffer_rsrc_t rsQu = make_rsrc_n(qu_base, qbytes), rsQv = make_rsrc_n(qv_base, qbytes),
                               rsDo = make_rsrc_n(hp.dout, (unsigned)(a.Tq - 1) * ostride + DK * 2u),
                               rsP = make_rsrc_n(REL ? hp.pos : hp.dout, REL ? (unsigned)(2 * a.Tq - 2) * pstride + DK * 2u : 0u);
  const int nstep = (a.Tq + 31) / 32;

  // ---- one-step-ahead operand fetch (whole block): registers now, LDS between the step's two barriers ------------------
  constexpr int RPP = NTHR / PER_ROW;
  static_assert(32 % RPP == 0, "a piece index must stay inside one 32-row operand tile");
  constexpr int NMAT = REL ? 3 : 2;  // 32-row tiles ahead of the band: Q+u, (Q+v,) dO
  Vec16<T> pre[PPT];
  float pre_lse = 0.f, pre_del = 0.f;
  bool pre_ok = false;
  unsigned pre_mask = 0xFFFFFFFFu;
  const bool has_mask = ws.mask != nullptr;
  const __amdgpu_buffer_rsrc_t rsL = make_rsrc_n(hp.lse, (unsigned)a.Tq * 4u), rsD = make_rsrc_n(hp.delta, (unsigned)a.Tq * 4u),
                               rsM = make_rsrc(ws.mask);
  static_assert(NTHR >= 32 * FW, "one mask word per thread");
  const int trow = tid / PER_ROW, piece = (tid % PER_ROW) * VEC;
  const unsigned q_lane = (unsigned)trow * qstride + (unsigned)piece * 2u, o_lane = (unsigned)trow * ostride + (unsigned)piece * 2u;
  const unsigned p_lane = (unsigned)((a.Tq - 32 + jblk + trow) * (int)pstride) + (unsigned)piece * 2u;   // band row t of step 0
  auto fetch = [&](const int step) {
    const int i0 = step * 32;
    if (wave == 0) {
      pre_ok = i0 + il < a.Tq;
      pre_lse = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsL, (unsigned)(i0 + il) * 4u, 0, 0));
      pre_del = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsD, (unsigned)(i0 + il) * 4u, 0, 0));
    }
    if (has_mask && tid < 32 * FW) {   // thread (wave slot tid >> 5, query tid & 31): word of key tile jblk / 32 + (tid >> 5)
      const int i = i0 + (tid & 31), w = jblk / 32 + (tid >> 5);
      const bool ok = i < a.Tq && w < ws.mask_nw;
      pre_mask = __builtin_amdgcn_raw_buffer_load_b32(rsM, ok ? (unsigned)(((((long)b * a.Tq + i) * a.H + h) * ws.mask_nw + w) * 4) : EMO_OOB, 0, 0);
    }
    const unsigned qo = q_lane + (unsigned)i0 * qstride, oo = o_lane + (unsigned)i0 * ostride;
    const unsigned po = p_lane - (unsigned)i0 * pstride;   // (wraps below row 0: out of range)
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      const int srow = p * RPP;
      if (srow < 32 * NMAT) {
        const int mat = srow / 32;
        const unsigned ro = (unsigned)(srow % 32);
        if (dbg & 4) pre[p].zero();
        else if (mat == 0) pre[p] = buf_load16<T>(rsQu, qo + ro * qstride);
        else if (REL && mat == 1) pre[p] = buf_load16<T>(rsQv, qo + ro * qstride);
        else pre[p] = buf_load16<T>(rsDo, oo + ro * ostride);
      } else {
        // the band block this step adds: rows Tq - 32 - i0 + jblk + [0, 32) (block -step of the ring)
        if (dbg & 1) pre[p].zero();
        else pre[p] = buf_load16<T>(rsP, po);
      }
    }
  };
  T* ring = stage0 + 96 * LD;   // (REL) blocks of 32 band rows: block n = rows Tq - 32 + jblk + 32 n + [0, 32) in slot n mod NRING
  auto ring_slot = [&](const int n) { return ((n % NRING) + NRING) % NRING; };
  auto stash = [&](const int step) {   // the tiles of `step` (fetched one step earlier)
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      if (p < NMAT) store16(stage0 + (32 * p + trow) * LD + piece, pre[p]);
      else store16(ring + (ring_slot(-step) * 32 + trow) * LD + piece, pre[p]);
    }
    if (wave == 0 && hh == 0) {
      // p = exp2(fma(s, c_exp, -lse * log2 e)): padding queries and rows without a valid key (lse = -inf) get -inf -> p = 0;
      // dS = p * fma(dP, keep * scale, -delta * scale): both row constants are stored negated (and delta scaled) for the fused multiply-adds
      rowc[il] = (pre_ok && pre_lse != -INFINITY) ? -pre_lse * 1.4426950408889634f : -INFINITY;
      rowc[32 + il] = -pre_del * a.scale;
    }
    if (tid < 32 * FW) maskw[tid] = pre_mask;   // (without dropout: all ones)
  };

  // ---- the workgroup's K and V tiles (rows jblk .. jblk + 32 FW; beyond Tk: zeros) --------------------------------------
  {
    const __amdgpu_buffer_rsrc_t rsK = make_rsrc(hp.k), rsV = make_rsrc(hp.v);
    T* kt0 = reinterpret_cast<T*>(smem + C_::kv_stage_bytes(REL));
    constexpr int KPT = 32 * FW * PER_ROW / NTHR;   // pieces per thread and operand (= 4)
    // every load of the prologue -- K, V, the band blocks 1 .. FW of step 0 (block 0 comes with fetch(0)) and the first step's
    // operands -- is issued before the first LDS store: ONE memory round trip ahead of the loop instead of three (the prologue and
    // epilogue were 30 % of a workgroup's time, in-kernel stamps of round 6)
    Vec16<T> kx[KPT], vx[KPT], rg[REL ? FW : 1];
#pragma unroll
    for (int p = 0; p < KPT; ++p) {
      const int pid = tid + NTHR * p, row = pid / PER_ROW, piece = (pid % PER_ROW) * VEC;
      const bool ok = jblk + row < a.Tk;
      kx[p] = buf_load16<T>(rsK, ok ? (unsigned)(((long)(jblk + row) * a.ldk + piece) * sizeof(T)) : EMO_OOB);
      vx[p] = buf_load16<T>(rsV, ok ? (unsigned)(((long)(jblk + row) * a.ldv + piece) * sizeof(T)) : EMO_OOB);
    }
    if constexpr (REL) {
#pragma unroll
      for (int n = 1; n <= FW; ++n) rg[n - 1] = (dbg & 1) ? Vec16<T>{} : buf_load16<T>(rsP, p_lane + (unsigned)(32 * n) * pstride);
    }
    fetch(0);
#pragma unroll
    for (int p = 0; p < KPT; ++p) {
      const int pid = tid + NTHR * p, row = pid / PER_ROW, piece = (pid % PER_ROW) * VEC;
      store16(kt0 + row * LD + piece, kx[p]);
      store16(kt0 + (32 * FW + row) * LD + piece, vx[p]);
    }
    if constexpr (REL) {
#pragma unroll
      for (int n = 1; n <= FW; ++n) store16(ring + (n * 32 + trow) * LD + piece, rg[n - 1]);
    }
  }
  const int kj = j0 + il;
  f32x16 dk[2], dv[2];
  zero16(dk[0]); zero16(dk[1]); zero16(dv[0]); zero16(dv[1]);
  float csum = 0.f;  // sum over this lane's queries of dS[:, key kj]
  stash(0);
  fetch(1);
  __syncthreads();
  const bool kvalid = kj < hp.klen;
  // a key past the utterance's length starts its score accumulator at -1e30: exp2 gives p = 0 exactly, and no select per element
  const float s_init = kvalid ? 0.f : -1e30f;
  // R6: this pass writes the dS image (bf16, query-major [b, h, i, ldds]) that the query pass (attn_bwd_q2_kernel) and the table
  // gradient read: the query pass no longer recomputes S, dP, the band product and the soft-max.  Lane = key: one 2-byte store per
  // accumulator register, the 32 lanes of a half wave cover 64 contiguous bytes; rows past Tq fall outside the descriptor.
  const bool has_img = REL && ws.dsq != nullptr && ws.img_from_kv;
  const __amdgpu_buffer_rsrc_t rsI = make_rsrc_n(has_img ? (const T*)ws.dsq + ((long)b * a.H + h) * a.Tq * ws.ldds : (const T*)hp.k,
                                                 has_img ? (unsigned)a.Tq * (unsigned)ws.ldds * 2u : 0u);
  const unsigned img_rstride = (unsigned)ws.ldds * 2u, img_lane = (unsigned)kj * 2u + (unsigned)(4 * hh) * img_rstride;
  const float c_exp = a.scale * 1.4426950408889634f;
  const unsigned keep_bits = __float_as_uint(a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f);
  const unsigned keep_bits_s = __float_as_uint((a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f) * a.scale);
  // the band skew as a lane rotation (ds_bpermute: no LDS storage): accumulator row rr + 4 hh of the band tiles G[query][c] gives
  // its element c = 31 - (rr + 4 hh) + il to lane il -- from tile 0 (c < 32, i.e. il <= row) or tile 1, the same source lane
  const int skew_base = il + 31 - 4 * hh;

  // this wave's K / V tile as MFMA operands, read from LDS once (round 6: the kernel is at 128 registers, the 32 these take no longer
  // push it past 256; they were 8 of the 40 16-byte LDS reads of a step)
  typename M_::Frag kf[NK], vf[NK];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    kf[kk] = M_::load_kc(Kt, LD, 0, kk * M_::KSTEP, lane);
    vf[kk] = M_::load_kc(Vt, LD, 0, kk * M_::KSTEP, lane);
  }
  EMO_WSTAMP(ws, 1);
  for (int step = 0; step < nstep; ++step) {
    const int i0 = step * 32;
    const T* Qus = stage0;
    const T* Qvs = stage0 + 32 * LD;
    const T* dOs = stage0 + (REL ? 64 : 32) * LD;
    // this wave's 64 band rows (REL): blocks wave - step and wave - step + 1 of the ring
    const T* Bs0 = ring + ring_slot(wave - step) * 32 * LD;
    const T* Bs1 = ring + ring_slot(wave - step + 1) * 32 * LD;
    EMO_STAMP(0);
    if (live) {
      // (Requesting every LDS operand of a phase ahead of its MFMAs -- 64 fragment registers in flight -- was measured: 373 against
      // 363 us at B 110, T' 320; the compiler's own placement, each read next to its MFMA, with the second wave of the SIMD
      // covering the round trip, stays.)
      f32x16 s, dp;
      zero16(dp);
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = s_init;
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        s = M_::mma(M_::load_kc(Qus, LD, 0, kk * M_::KSTEP, lane), kf[kk], s);
        dp = M_::mma(M_::load_kc(dOs, LD, 0, kk * M_::KSTEP, lane), vf[kk], dp);
      }
      EMO_STAMP(1);
      if constexpr (REL) {
        // band product G[query][band column c] = (Q+v) band^T, two 32-column tiles
        f32x16 g0, g1;
        zero16(g0); zero16(g1);
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
          const typename M_::Frag fv = M_::load_kc(Qvs, LD, 0, kk * M_::KSTEP, lane);
          g0 = M_::mma(fv, M_::load_kc(Bs0, LD, 0, kk * M_::KSTEP, lane), g0);
          g1 = M_::mma(fv, M_::load_kc(Bs1, LD, 0, kk * M_::KSTEP, lane), g1);
        }
        EMO_STAMP(2);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = (r & 3) + 8 * (r >> 2);
          const int addr = (((skew_base - rr) & 31) + 32 * hh) * 4;
          // the choice between the two band tiles is made on the SOURCE lane: lane c of row q holds band column c of tile 0 and 32 + c
          // of tile 1, and the one reader of that lane and row wants tile 0 exactly when c >= 31 - q -- one rotation per element, not two
          const float gv = (il >= 31 - (rr + 4 * hh)) ? g0[r] : g1[r];
          s[r] += (dbg & 2) ? gv : __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(gv)));
        }
      }
      EMO_STAMP(3);
      // soft-max, dropout, dS: accumulator rows 4 g .. 4 g + 3 are queries i0 + 8 g + 4 hh + 0 .. 3
      typename M_::Frag pf[NS], df[NS];   // P and dS as the next products' B operands (accumulator rows 8 ks .. 8 ks + 7, chain_b's order)
      unsigned dsw[8];                    // dS of rows (8 g + 4 hh + 2 h2, + 1) as bf16 pairs: word 2 g + h2
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(rowc + 8 * g + 4 * hh);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(rowc + 32 + 8 * g + 4 * hh);
        const u32x4 m4 = *reinterpret_cast<const u32x4*>(maskw + 32 * wave + 8 * g + 4 * hh);   // keep bits of the rows' keys j0 .. j0 + 31
        float dsf[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c_exp, l4[e]));
          // bit il of the row's word: 0 / -1 -> 0 / keep (attn_dropmask_kernel hashed the mask once for both passes)
          const unsigned kb = (unsigned)__builtin_amdgcn_sbfe(m4[e], il, 1);
          const float m = __uint_as_float(kb & keep_bits);
          const float dsv = p * __builtin_fmaf(dp[r], __uint_as_float(kb & keep_bits_s), d4[e]);
          csum += dsv;
          pf[g >> 1][4 * (g & 1) + e] = (bf16)(p * m);
          dsf[e] = dsv;
        }
        // dS as explicit bf16 PAIRS: one conversion per pair serves the operand register and both image stores (low / high half)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          bf16x2 pr;
          pr[0] = (bf16)dsf[2 * h2];
          pr[1] = (bf16)dsf[2 * h2 + 1];
          dsw[2 * g + h2] = __builtin_bit_cast(unsigned, pr);
        }
      }
#pragma unroll
      for (int ks = 0; ks < NS; ++ks)
        df[ks] = __builtin_bit_cast(typename M_::Frag, u32x4{dsw[4 * ks], dsw[4 * ks + 1], dsw[4 * ks + 2], dsw[4 * ks + 3]});
      // the image rows, from the PACKED words (low / high halves: no second conversion per element; extracting the elements of the
      // operand vector with __builtin_bit_cast compiled into 16 stores of two registers -- a wrong image).  UNCONDITIONAL: without
      // an image the descriptor has no records and the store falls outside it (an `if (has_img)` here was a scalar branch per element:
      // 16 basic blocks per step that nothing could be scheduled across).  The row's offset is the scalar operand.  (Pairing two lanes'
      // values into 4-byte stores over DPP was measured: 110 against 103 us at B 110, T' 320.)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(dsw[2 * g + (e >> 1)] >> (16 * (e & 1))), rsI, img_lane,
                                                (unsigned)(i0 + 8 * g + e) * img_rstride, 0);
      EMO_STAMP(4);
      // dV^T += dO^T P,  dK^T += (Q+u)^T dS: the accumulators are the B operands (query index in the registers)
#pragma unroll
      for (int ks = 0; ks < NS; ++ks) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dv[dt] = M_::mma(chain_a<T, TR>(dOs, ks, 32 * dt, lane), pf[ks], dv[dt]);
          dk[dt] = M_::mma(chain_a<T, TR>(Qus, ks, 32 * dt, lane), df[ks], dk[dt]);
        }
      }
    }
    EMO_STAMP(5);
    lds_barrier();   // every wave has read the stage
    EMO_STAMP(6);
    stash(step + 1);   // tiles of step + 1 (fetched during the previous step)
    EMO_STAMP(7);
    fetch(step + 2);
    EMO_STAMP(8);
    lds_barrier();   // stage ready
    EMO_STAMP(9);
  }
  EMO_WSTAMP(ws, 2);
  store_dT<T>((T*)hp.dk, a.ldk, j0, a.Tk, dk, 1.f, lane);  // (a dead wave stores zeros)
  store_dT<T>((T*)hp.dv, a.ldv, j0, a.Tk, dv, 1.f, lane);
  if (REL && a.dbias_u && live) {
    // dbias_u[d] += sum_j colsum_j K[j][d];  dbias_v[d] -= that (attn_bwd_q_kernel adds colsum(dQ): dbias_v = colsum(dQ) - dbias_u)
    csum += __shfl_xor(csum, 32, 64);
    float acc = 0.f;
#pragma unroll 8
    for (int j = 0; j < 32; ++j) acc += __shfl(csum, j, 64) * (float)Kt[j * LD + lane];
    atomicAdd(&a.dbias_u[h * DK + lane], acc);
    atomicAdd(&a.dbias_v[h * DK + lane], -acc);
  }
  EMO_WSTAMP(ws, 3);
}

template <typename T, bool TR, bool REL, int FW>
__global__ __launch_bounds__(64 * FW, 2) void attn_bwd_q_kernel(const emoasr_attn_t a_in, const FusedWs ws_in, const int nt) {
  using M_ = Mma<T>;
  using C_ = SplitCfg<T, FW>;
  constexpr int NK = AttnCfg<T>::NK, NS = AttnCfg<T>::NS, LD = C_::LD, LDG = C_::LDG;
  constexpr int VEC = 16 / sizeof(T), PER_ROW = DK / VEC;
  constexpr int NTHR = 64 * FW;
  constexpr int PPT = REL ? 3 : 2, NRING = C_::RING_BLOCKS;   // K, V and (relative positions) the ONE band block a step adds: the
  static_assert(NTHR / PER_ROW == 32, "one piece per thread covers one 32-row tile");   // band is a ring (see attn_bwd_kv_kernel)
  static_assert(sizeof(T) == 2, "the two-pass backward is bf16 only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, il = lane & 31, hh = lane >> 5;
  emoasr_attn_t a = a_in;
  FusedWs ws = ws_in;
  const Blk3 blk = attn_block(nt, a_in.H, a_in.B);
  if (!blk.ok) return;
  int b = blk.z;
  if (a_in.nseg > 1) {
    const SegRef g = seg_of_slot(a_in, blk.z, b);
    seg_apply<T>(a, g);
    ws.qu = (const T*)ws.qu + g.row * ws.ldqu;
    ws.qv = (const T*)ws.qv + g.row * ws.ldqu;
    if (ws.dsq) ws.dsq = (T*)ws.dsq + g.row * a.H * ws.ldds;
    if (ws.mask) ws.mask += g.row * a.H * ws.mask_nw;
  }
  const int iblk = blk.x * (32 * FW), h = blk.y;
  if (iblk >= a.Tq) return;   // (a shorter segment of a stacked launch: the grid follows the longest)
  const HeadPtrs hp = head_ptrs<T>(a, b, h);
  const int i0 = iblk + 32 * wave;
  const bool live = i0 < a.Tq;   // a dead wave still stages and joins the barriers
  const int qi = i0 + il;
  const bool qval = qi < a.Tq;

  T* stage0 = reinterpret_cast<T*>(smem);
  char* mine = smem + C_::q_stage_bytes(REL) + wave * C_::wave_bytes(REL);
  float* Gs = reinterpret_cast<float*>(mine);   // G[32 queries][LDGS band columns] f32 (16-byte stores of four band columns) ...
  constexpr int LDGS = C_::LDGS;
  T* img_g = reinterpret_cast<T*>(mine);        // ... then dG[query][band column] ([32][LDG]), read back k-contiguous

  const long ho = (long)h * DK;
  const T* qu_base = (const T*)ws.qu + (long)b * a.Tq * ws.ldqu + ho;
  const T* qv_base = REL ? (const T*)ws.qv + (long)b * a.Tq * ws.ldqu + ho : qu_base;
  // descriptors sized to the valid rows (keys < Tk, table rows < 2 Tq - 1): see attn_fwd_kernel
  const unsigned kstride = (unsigned)a.ldk * 2u, vstride = (unsigned)a.ldv * 2u, pstride = (unsigned)a.ldp * 2u;
  const __amdgpu_buffer_rsrc_t rsK = make_rsrc_n(hp.k, (unsigned)(a.Tk - 1) * kstride + DK * 2u),
                               rsV = make_rsrc_n(hp.v, (unsigned)(a.Tk - 1) * vstride + DK * 2u),
                               rsP = make_rsrc_n(REL ? hp.pos : hp.k, REL ? (unsigned)(2 * a.Tq - 2) * pstride + DK * 2u : 0u);
  const int nstep = (hp.klen + 31) / 32;   // key tiles with at least one valid key

  Vec16<T> pre[PPT];
  T* ring = stage0 + 64 * LD;   // (REL) block n = table rows Tq - 32 FW - iblk + 32 n + [0, 32) in slot n mod NRING
  const int trow = tid / PER_ROW, piece = (tid % PER_ROW) * VEC;
  const unsigned k_lane = (unsigned)trow * kstride + (unsigned)piece * 2u, v_lane = (unsigned)trow * vstride + (unsigned)piece * 2u;
  // band rows of the block's 32 FW queries against key tile j0: r = Tq - 32 FW - iblk + j0 + t, t in [0, 32 FW + 32)
  const unsigned p_lane = (unsigned)((a.Tq - 32 * FW - iblk + trow) * (int)pstride) + (unsigned)piece * 2u;
  auto fetch = [&](const int step) {
    const int j0 = step * 32;
    const unsigned dead = step < nstep ? 0u : 0x80000000u;   // behind the last key tile: nothing is read
    const unsigned ko = (k_lane + (unsigned)j0 * kstride) | dead, vo = (v_lane + (unsigned)j0 * vstride) | dead;
    pre[0] = buf_load16<T>(rsK, ko);
    pre[1] = buf_load16<T>(rsV, vo);
    if constexpr (REL) pre[2] = buf_load16<T>(rsP, (p_lane + (unsigned)(j0 + 32 * FW) * pstride) | dead);   // block step + FW
  };
  auto stash = [&](const int step) {
    store16(stage0 + trow * LD + piece, pre[0]);
    store16(stage0 + (32 + trow) * LD + piece, pre[1]);
    if constexpr (REL) store16(ring + (((step + FW) % NRING) * 32 + trow) * LD + piece, pre[2]);
  };

  // ---- stationary operands of this wave's query tile (B operands: query on the lane) ---------------------------
  typename M_::Frag fqu[NK], fqv[NK], fdo[NK];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    fqu[kk] = frag_global<T>(qu_base, ws.ldqu, qi, live && qval, kk, lane, nullptr);
    if constexpr (REL) fqv[kk] = frag_global<T>(qv_base, ws.ldqu, qi, live && qval, kk, lane, nullptr);
    fdo[kk] = frag_global<T>((const T*)hp.dout, a.ldo, qi, live && qval, kk, lane, nullptr);
  }
  float lse2 = INFINITY, del_q = 0.f;
  if (live && qval) {
    const float l = hp.lse[qi];
    lse2 = l != -INFINITY ? l * 1.4426950408889634f : INFINITY;
    del_q = hp.delta[qi];
  }
  f32x16 dq[2];
  zero16(dq[0]); zero16(dq[1]);
  if constexpr (REL) {   // blocks 0 .. FW - 1 of step 0 (block FW comes with fetch(0))
#pragma unroll
    for (int n = 0; n < FW; ++n) store16(ring + (n * 32 + trow) * LD + piece, buf_load16<T>(rsP, p_lane + (unsigned)(32 * n) * pstride));
  }
  fetch(0);
  stash(0);
  fetch(1);
  __syncthreads();
  const float c_exp = a.scale * 1.4426950408889634f;
  const unsigned keep_bits = __float_as_uint(a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f);
  // the keep-mask word of (query qi, key tile `step`), one step ahead (attn_dropmask_kernel; without dropout: all ones)
  const bool has_mask = ws.mask != nullptr;
  const __amdgpu_buffer_rsrc_t rsM = make_rsrc(ws.mask);
  const unsigned mrow = (unsigned)((((long)b * a.Tq + qi) * a.H + h) * ws.mask_nw * 4);
  auto mask_word = [&](const int step) -> unsigned {
    if (!has_mask) return 0xFFFFFFFFu;
    return __builtin_amdgcn_raw_buffer_load_b32(rsM, (live && qval && step < nstep) ? mrow + 4u * (unsigned)step : EMO_OOB, 0, 0);
  };
  unsigned mw_next = mask_word(0);

  for (int step = 0; step < nstep; ++step) {
    const int j0 = step * 32;
    const T* Ks = stage0;
    const T* Vs = stage0 + 32 * LD;
    // this wave's 64 band rows (REL): blocks FW - 1 - wave + step and the next
    const T* Bs0 = ring + ((FW - 1 - wave + step) % NRING) * 32 * LD;
    const T* Bs1 = ring + ((FW - wave + step) % NRING) * 32 * LD;
    const unsigned mw = mw_next >> (4 * hh);   // this half wave's keys are 8 g + 4 hh + e
    mw_next = mask_word(step + 1);
    EMO_STAMP(0);
    if (live) {
      f32x16 s, dp;
      zero16(s); zero16(dp);
      if constexpr (REL) {
        f32x16 g0, g1;
        zero16(g0); zero16(g1);
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
          g0 = M_::mma(M_::load_kc(Bs0, LD, 0, kk * M_::KSTEP, lane), fqv[kk], g0);   // g[c][i]
          g1 = M_::mma(M_::load_kc(Bs1, LD, 0, kk * M_::KSTEP, lane), fqv[kk], g1);
        }
        // (S and dP are issued BEFORE the band tiles are stored: the stores wait for the band MFMAs' results while the matrix pipe
        // works on the two other products)
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
          s = M_::mma(M_::load_kc(Ks, LD, 0, kk * M_::KSTEP, lane), fqu[kk], s);
          dp = M_::mma(M_::load_kc(Vs, LD, 0, kk * M_::KSTEP, lane), fdo[kk], dp);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {   // accumulator registers 4 q .. 4 q + 3 are band columns 8 q + 4 hh + 0 .. 3 of query il
          *reinterpret_cast<f32x4*>(Gs + il * LDGS + 8 * q + 4 * hh) = f32x4{g0[4 * q], g0[4 * q + 1], g0[4 * q + 2], g0[4 * q + 3]};
          *reinterpret_cast<f32x4*>(Gs + il * LDGS + 32 + 8 * q + 4 * hh) = f32x4{g1[4 * q], g1[4 * q + 1], g1[4 * q + 2], g1[4 * q + 3]};
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
          s = M_::mma(M_::load_kc(Ks, LD, 0, kk * M_::KSTEP, lane), fqu[kk], s);
          dp = M_::mma(M_::load_kc(Vs, LD, 0, kk * M_::KSTEP, lane), fdo[kk], dp);
        }
      }
      EMO_STAMP(1);
      if constexpr (REL) {
        __builtin_amdgcn_wave_barrier();
        // element (key jl, query il) sits in band column c = 31 - il + jl
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] += Gs[il * (LDGS - 1) + 31 + c_row(r, lane)];   // G[il][31 - il + key]
        __builtin_amdgcn_wave_barrier();
        // the dG image takes the skew tile's place: cleared with 16-byte stores, the tile's 32 x 32 entries written below
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int v = lane + 64 * q;   // 32 rows x 8 pieces of 8 columns
          *reinterpret_cast<u32x4*>(img_g + (v >> 3) * LDG + 8 * (v & 7)) = u32x4{0u, 0u, 0u, 0u};
        }
        __builtin_amdgcn_wave_barrier();
      }
      EMO_STAMP(2);
      // soft-max, dropout, dS -- packed straight into the B operand of the dQ product (accumulator rows 8 ks .. 8 ks + 7, chain_b's
      // order); accumulator rows 4 g .. 4 g + 3 are keys j0 + 8 g + 4 hh + 0 .. 3: two hash pairs (drop_base, j0 even)
      typename M_::Frag df[NS];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          float p = __builtin_amdgcn_exp2f(s[r] * c_exp - lse2);
          p = (j0 + c_row(r, lane) >= hp.klen) ? 0.f : p;
          // bit 8 g + e of the half wave's word: 0 / -1 -> 0 / keep
          const float m = __uint_as_float((unsigned)__builtin_amdgcn_sbfe(mw, 8 * g + e, 1) & keep_bits);
          const bf16 dsv = (bf16)(p * (dp[r] * m - del_q) * a.scale);
          df[g >> 1][4 * (g & 1) + e] = dsv;
          if constexpr (REL) {
            // dG[query il][band column c = key - il + 31] (the image was cleared after the skew read)
            img_g[il * LDG + c_row(r, lane) + 31 - il] = dsv;
          }
        }
      }
      EMO_STAMP(3);
      if constexpr (REL) {
        // dS for the position-table gradient: query-major rows, 4 consecutive keys (8 bytes) per store
        if (qval) {
          T* drow = (T*)ws.dsq + (((long)b * a.H + h) * a.Tq + qi) * ws.ldds + j0 + 4 * hh;
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            bf16x4 v4;
#pragma unroll
            for (int e = 0; e < 4; ++e) v4[e] = df[g4 >> 1][4 * (g4 & 1) + e];
            *reinterpret_cast<bf16x4*>(drow + 8 * g4) = v4;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      EMO_STAMP(4);
      // dQ^T += K^T dS^T (the accumulator is the B operand: key index in the registers) + band^T unskew(dS^T)
#pragma unroll
      for (int ks = 0; ks < NS; ++ks) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) dq[dt] = M_::mma(chain_a<T, TR>(Ks, ks, 32 * dt, lane), df[ks], dq[dt]);
      }
      if constexpr (REL) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int ks = 0; ks < NS; ++ks) {
            const typename M_::Frag dg = M_::load_kc(img_g, LDG, 0, 32 * ct + ks * M_::KSTEP, lane);  // dG^T[c][i], c contiguous
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
              dq[dt] = M_::mma(M_::template load_km<TR>(ct ? Bs1 : Bs0, LD, ks * M_::KSTEP, 32 * dt, lane), dg, dq[dt]);
          }
        __builtin_amdgcn_wave_barrier();   // the image has been read: the region takes the next step's skew tile
      }
    }
    EMO_STAMP(5);
    lds_barrier();   // every wave has read the stage
    EMO_STAMP(6);
    stash(step + 1);
    EMO_STAMP(7);
    fetch(step + 2);
    EMO_STAMP(8);
    lds_barrier();   // stage ready
    EMO_STAMP(9);
  }
  if (live) store_dT<T>((T*)hp.dq, a.ldq, i0, a.Tq, dq, 1.f, lane);
  if (REL && a.dbias_v && live) {
    // dbias_v[d] += colsum(dQ)[d] (attn_bwd_kv_kernel subtracts dbias_u): sum over the queries = the 32 lanes of a half wave
    float* cs = Gs;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = dq[dt][r];
        v = half_sum32(v);
        if (il == 31) cs[32 * dt + c_row(r, lane)] = v;
      }
    __builtin_amdgcn_wave_barrier();
    atomicAdd(&a.dbias_v[h * DK + lane], cs[lane]);
  }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Query pass, round 6 (bf16, relative positions): dQ from the dS image the key pass wrote.  attn_bwd_q_kernel recomputed every score
// tile a second time -- band product (8 MFMAs), S and dP (8), skew through LDS, soft-max, mask -- only to arrive at the dS the key
// pass had just formed; 16 of its 28 MFMAs and ~200 of its ~250 VALU instructions per tile were recomputation.  Here a wave reads
// its query tile's dS rows back (four 8-byte loads per lane and key tile, one step ahead: exactly the values, rounded to bf16, that
// the old pass fed its MFMAs) and does what only this orientation can: dQ^T += K^T dS^T, and the band part through the un-skewed
// dG image -- 12 MFMAs per tile, no exponentials.  LDS: the K tile + the band ring (32 KB) + a 4.6 KB image per wave: three
// workgroups per CU; ~100 registers.
// ------------------------------------------------------------------------------------------------------------------------------------
template <int FW> struct Q2Cfg {
  static constexpr int LD = AttnCfg<bf16>::LD, LDG = 72, NRING = FW + 2;
  static constexpr int STAGE_ROWS = 32 + 32 * NRING;   // K, band ring
  static constexpr int IMG_BYTES = 32 * LDG * 2;
  static constexpr int smem() { return STAGE_ROWS * LD * 2 + FW * IMG_BYTES; }
};
template <bool TR, int FW>
__global__ __launch_bounds__(64 * FW, 3) void attn_bwd_q2_kernel(const emoasr_attn_t a_in, const FusedWs ws_in, const int nt) {
  using T = bf16;
  using M_ = Mma<T>;
  using C_ = Q2Cfg<FW>;
  constexpr int NS = AttnCfg<T>::NS, LD = C_::LD, LDG = C_::LDG, NRING = C_::NRING;
  constexpr int VEC = 8, PER_ROW = DK / VEC, NTHR = 64 * FW;
  static_assert(NTHR / PER_ROW == 32, "one piece per thread covers one 32-row tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), il = lane & 31, hh = lane >> 5;
  emoasr_attn_t a = a_in;
  FusedWs ws = ws_in;
  EMO_WSTAMP(ws_in, 0);
  const Blk3 blk = attn_block(nt, a_in.H, a_in.B);
  if (!blk.ok) return;
  int b = blk.z;
  if (a_in.nseg > 1) {
    const SegRef g = seg_of_slot(a_in, blk.z, b);
    seg_apply<T>(a, g);
    ws.dsq = (T*)ws.dsq + g.row * a.H * ws.ldds;
  }
  const int iblk = blk.x * (32 * FW), h = blk.y;
  if (iblk >= a.Tq) return;
  const HeadPtrs hp = head_ptrs<T>(a, b, h);
  const int i0 = iblk + 32 * wave;
  const bool live = i0 < a.Tq;   // a dead wave still stages and joins the barriers
  const int qi = i0 + il;
  const bool qval = qi < a.Tq;

  T* Ks = reinterpret_cast<T*>(smem);
  T* ring = Ks + 32 * LD;   // block n = table rows Tq - 32 FW - iblk + 32 n + [0, 32) in slot n mod NRING
  T* img_g = reinterpret_cast<T*>(smem + C_::STAGE_ROWS * LD * 2 + wave * C_::IMG_BYTES);   // dG[query][band column] ([32][LDG])

  const int nstep = (hp.klen + 31) / 32;   // key tiles with at least one valid key (the only ones the key pass wrote)
  const unsigned kstride = (unsigned)a.ldk * 2u, pstride = (unsigned)a.ldp * 2u, istride = (unsigned)ws.ldds * 2u;
  const __amdgpu_buffer_rsrc_t rsK = make_rsrc_n(hp.k, (unsigned)(a.Tk - 1) * kstride + DK * 2u),
                               rsP = make_rsrc_n(hp.pos, (unsigned)(2 * a.Tq - 2) * pstride + DK * 2u),
                               rsI = make_rsrc_n((const T*)ws.dsq + ((long)b * a.H + h) * a.Tq * ws.ldds, (unsigned)a.Tq * istride);
  const int trow = tid / PER_ROW, piece = (tid % PER_ROW) * VEC;
  const unsigned k_lane = (unsigned)trow * kstride + (unsigned)piece * 2u;
  const unsigned p_lane = (unsigned)((a.Tq - 32 * FW - iblk + trow) * (int)pstride) + (unsigned)piece * 2u;   // block 0
  // K tile of `step` and the band block it adds (block step + FW): THREE named register sets, tile t in set t mod 3, so a load
  // has three steps to land before its stash (with one step of lead a workgroup's time was nstep x the memory round trip:
  // 12 MFMAs per tile are far shorter than that)
  Vec16<T> pA[2], pB[2], pC[2];
  auto fetch = [&](Vec16<T> (&pre)[2], const int step) {
    const unsigned dead = step < nstep ? 0u : 0x80000000u;
    pre[0] = buf_load16<T>(rsK, (k_lane + (unsigned)(32 * step) * kstride) | dead);
    pre[1] = buf_load16<T>(rsP, (p_lane + (unsigned)(32 * (step + FW)) * pstride) | dead);
  };
  auto stash = [&](const Vec16<T> (&pre)[2], const int step) {
    store16(Ks + trow * LD + piece, pre[0]);
    store16(ring + (((step + FW) % NRING) * 32 + trow) * LD + piece, pre[1]);
  };
  // this lane's dS values of key tile `step`: row qi, keys 32 step + 8 g + 4 hh + 0 .. 3 (the B-operand order of the dQ product)
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
  const unsigned i_lane = (unsigned)qi * istride + (unsigned)(4 * hh) * 2u;
  auto fetch_ds = [&](u32x2 (&d)[4], const int step) {
    const unsigned dead = (live && qval && step < nstep) ? 0u : 0x80000000u;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const unsigned off = (i_lane + (unsigned)(32 * step + 8 * g) * 2u) | dead;
      d[g] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsI, off, 0, 0));
    }
  };
  {
    Vec16<T> rg[FW];
#pragma unroll
    for (int n = 0; n < FW; ++n) rg[n] = buf_load16<T>(rsP, p_lane + (unsigned)(32 * n) * pstride);
    fetch(pA, 0);
#pragma unroll
    for (int n = 0; n < FW; ++n) store16(ring + (n * 32 + trow) * LD + piece, rg[n]);
  }
  stash(pA, 0);
  fetch(pB, 1);
  fetch(pC, 2);
  fetch(pA, 3);
  // the dS rows come straight from global memory into registers, THREE key tiles ahead (three named register sets: with one tile
  // of lead the pass waited a memory round trip per step -- 12 MFMAs per tile leave nothing to hide it behind)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int v = lane + 64 * q;   // 32 rows x 8 pieces of 8 columns
    *reinterpret_cast<u32x4*>(img_g + (v >> 3) * LDG + 8 * (v & 7)) = u32x4{0u, 0u, 0u, 0u};
  }
  u32x2 dr0[4], dr1[4], dr2[4];
  fetch_ds(dr0, 0);
  fetch_ds(dr1, 1);
  fetch_ds(dr2, 2);
  __syncthreads();
  f32x16 dq[2];
  zero16(dq[0]); zero16(dq[1]);
  typename M_::Frag bcar[NS][2];   // fragments of this wave's lower band block of the coming step (block FW - 1 - wave + step)
  if (live) {
#pragma unroll
    for (int ks = 0; ks < NS; ++ks)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) bcar[ks][dt] = M_::template load_km<TR>(ring + ((FW - 1 - wave) % NRING) * 32 * LD, LD, ks * M_::KSTEP, 32 * dt, lane);
  }

  auto body = [&](const int step, u32x2 (&dset)[4], Vec16<T> (&pset)[2]) __attribute__((always_inline)) {
    u32x2 dcur[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) dcur[g] = dset[g];
    EMO_STAMP(0);
    fetch_ds(dset, step + 3);
    if (live) {
      const T* Bs1 = ring + ((FW - wave + step) % NRING) * 32 * LD;
      // the dG image: the tile's 32 x 32 entries at band column c = key - il + 31 -- the same entries at every step, so the rest of the
      // image was cleared once, ahead of the loop
      typename M_::Frag df[NS];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        union { u32x2 u; bf16x4 h; } cv;
        cv.u = dcur[g];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          df[g >> 1][4 * (g & 1) + e] = cv.h[e];
          img_g[il * LDG + 8 * g + 4 * hh + e + 31 - il] = cv.h[e];
        }
      }
      __builtin_amdgcn_wave_barrier();
      EMO_STAMP(1);
      // dQ^T += K^T dS^T (key index in the registers) + band^T unskew(dS^T)
#pragma unroll
      for (int ks = 0; ks < NS; ++ks) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) dq[dt] = M_::mma(chain_a<T, TR>(Ks, ks, 32 * dt, lane), df[ks], dq[dt]);
      }
      EMO_STAMP(2);
      // the band operands: this step's upper block (Bs1) is the next step's lower one (Bs0) -- its fragments stay in registers
      // (`bcar`), so a step reads ONE block's fragments from LDS, not two
      typename M_::Frag bnew[NS][2];
#pragma unroll
      for (int ks = 0; ks < NS; ++ks)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) bnew[ks][dt] = M_::template load_km<TR>(Bs1, LD, ks * M_::KSTEP, 32 * dt, lane);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int ks = 0; ks < NS; ++ks) {
          const typename M_::Frag dg = M_::load_kc(img_g, LDG, 0, 32 * ct + ks * M_::KSTEP, lane);  // dG^T[c][i], c contiguous
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) dq[dt] = M_::mma(ct ? bnew[ks][dt] : bcar[ks][dt], dg, dq[dt]);
        }
#pragma unroll
      for (int ks = 0; ks < NS; ++ks)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) bcar[ks][dt] = bnew[ks][dt];
      __builtin_amdgcn_wave_barrier();   // the image has been read
    }
    EMO_STAMP(3); EMO_STAMP(4); EMO_STAMP(5);
    lds_barrier();   // every wave has read the stage
    EMO_STAMP(6);
    stash(pset, step + 1);
    EMO_STAMP(7);
    fetch(pset, step + 4);
    EMO_STAMP(8);
    lds_barrier();   // stage ready
    EMO_STAMP(9);
  };
  EMO_WSTAMP(ws, 1);
  for (int step = 0; step < nstep; step += 3) {
    body(step, dr0, pB);
    if (step + 1 < nstep) body(step + 1, dr1, pC);
    if (step + 2 < nstep) body(step + 2, dr2, pA);
  }
  EMO_WSTAMP(ws, 2);
  if (!live) return;
  store_dT<T>((T*)hp.dq, a.ldq, i0, a.Tq, dq, 1.f, lane);
  if (a.dbias_v) {
    // dbias_v[d] += colsum(dQ)[d] (attn_bwd_kv_kernel subtracts dbias_u): sum over the queries = the 32 lanes of a half wave
    float* cs = reinterpret_cast<float*>(img_g);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = dq[dt][r];
        v = half_sum32(v);
        if (il == 31) cs[32 * dt + c_row(r, lane)] = v;
      }
    __builtin_amdgcn_wave_barrier();
    atomicAdd(&a.dbias_v[h * DK + lane], cs[lane]);
  }
  EMO_WSTAMP(ws, 3);
}

template <typename T>
__global__ __launch_bounds__(256) void attn_cast_kernel(const float* __restrict__ src, T* __restrict__ dst, const long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = from_f32<T>(src[i]);
}

// dq (T, strided) = sum over the utterance's live key blocks of their dQ partials, in block order (bit-reproducible).
// Block = 16 rows x ncol columns; the slab loads of a row are issued together (out-of-range blocks: no traffic).
struct FinSegs { int n; int b0[EMOASR_MAX_SEGMENTS + 1], T[EMOASR_MAX_SEGMENTS]; long row[EMOASR_MAX_SEGMENTS + 1]; };
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_fin_kernel(const long rows, const int ncol, const float* __restrict__ dq32,
                                                           const long slab, const int keys_per_block, const int Tq, const int Tk,
                                                           const int* __restrict__ klens, T* __restrict__ dq, const long ldq,
                                                           const float* __restrict__ cast_src, T* __restrict__ cast_dst,
                                                           const long cast_n, const FinSegs sg) {
  // on the side: the finished f32 position-table gradient rounded to the compute dtype for its weight-gradient product
  // (this launch follows attn_bwd_dpos3_kernel; saves the separate cast launch of the per-layer backward)
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < cast_n; i += (long)gridDim.x * 256) cast_dst[i] = from_f32<T>(cast_src[i]);
  const int ngrp = ncol / 8, rpb = 256 / ngrp;  // column groups of 8, rows per pass
  const int cg = threadIdx.x % ngrp, rr = threadIdx.x / ngrp;
  if (rr >= rpb) return;
  const long row_lo = (long)blockIdx.x * 16, row_hi = min(rows, row_lo + 16);
  for (long row = row_lo + rr; row < row_hi; row += rpb) {
    int b = (int)(row / Tq), tk = Tk;
    if (sg.n > 1) {   // stacked micro-batches: the row's segment gives its utterance and padded length
      int si = 0;
      for (int k = 1; k < EMOASR_MAX_SEGMENTS; ++k) si += (k < sg.n && row >= sg.row[k]) ? 1 : 0;
      tk = sg.T[si];
      b = sg.b0[si] + (int)((row - sg.row[si]) / tk);
    }
    const int klen = klens ? min(klens[b], tk) : tk;
    const int nkb = (klen + keys_per_block - 1) / keys_per_block;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    const int nkb_max = (Tk + keys_per_block - 1) / keys_per_block;  // uniform loop bound
    for (int kb0 = 0; kb0 < nkb_max; kb0 += 4) {
      f32x4 v[4][2];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool on = kb0 + u < nkb;  // (nkb varies with the row's utterance: a per-lane predicate, not a branch)
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(dq32 + (long)(kb0 + u) * slab);  // wave-uniform base
        const unsigned off = (unsigned)((row * ncol + cg * 8) * 4);
        v[u][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, on ? off : EMO_OOB, 0, 0));
        v[u][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, on ? off + 16u : EMO_OOB, 0, 0));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] += v[u][0][e]; o[4 + e] += v[u][1][e]; }
    }
    store8<T>(dq + row * ldq + cg * 8, o);
  }
}

template <typename K>
int set_smem(K kernel, int bytes) {
  if (bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { emo_set_error("hipFuncSetAttribute(%d): %s", bytes, hipGetErrorString(e)); return 1; }
  }
  return 0;
}

int g_tr = 1;
int g_q2 = 1;         // option "attn_q2": the query pass reads the dS image the key pass wrote (attn_bwd_q2_kernel) instead of recomputing it
int g_bwd_split = 1;  // option "attn_bwd_split": the two-pass backward (attn_bwd_kv_kernel + attn_bwd_q_kernel); 0 = the single-pass kernel
int g_fused_fw = 0;  // key tiles per workgroup of the single-pass backward (0 = by grid size; emoasr_set_option "attn_fw")

int g_fwd4 = 1;       // option "attn_fwd4": the block-staged forward (attn_fwd4_kernel) for the training launches
int g_fwd_split = 1;  // option "attn_fwd_split": key split for small launches
int g_fwd_waves = 0;  // option "attn_fwd_waves": waves per workgroup of attn_fwd_kernel (0 = 1 for stacked launches, else 4)
int g_attn_xcd = 1;   // option "attn_xcd": stacked forward / fused-backward launches keep a (head, utterance) group on one XCD
int g_attn_lpt = 1;   // option "attn_lpt": 0 = segments in stacking order
// dispatch order of a stacked launch's segments: longest first (stable), see seg_of_slot
void fill_seg_order(emoasr_attn_t& a) {
  for (int k = 0; k < EMOASR_MAX_SEGMENTS; ++k) a.seg_order[k] = k;
  if (a.nseg > 1 && g_attn_lpt)
    std::stable_sort(a.seg_order, a.seg_order + a.nseg, [&](int x, int y) { return a.seg_T[x] > a.seg_T[y]; });
}
int check_args(const emoasr_attn_t* a, int dtype) {
  EMO_CHECK(a->DK == DK, "attention: DK=%d unsupported (64 only)", a->DK);
  const int vec = dtype == EMO_BF16 ? 8 : 4;
  EMO_CHECK(a->ldq % vec == 0 && a->ldk % vec == 0 && a->ldv % vec == 0 && a->ldo % vec == 0,
            "attention: row strides must be multiples of %d elements", vec);
  EMO_CHECK(!a->pos || (a->Tq == a->Tk && a->ldp % vec == 0), "attention: relative positions need Tq == Tk");
  EMO_CHECK(!(a->pos && a->causal), "attention: causal + relative positions unsupported");
  if (a->nseg > 1) {   // stacked micro-batches
    EMO_CHECK(a->nseg <= EMOASR_MAX_SEGMENTS && a->Tq == a->Tk && !a->causal && !a->st && !a->pdT,
              "attention: stacked segments need self-attention without causal mask / stored scores / materialised scratch");
    EMO_CHECK(a->seg_b0[0] == 0 && a->seg_row[0] == 0 && a->seg_b0[a->nseg] == a->B, "attention: bad segment table");
    int tmax = 0;
    for (int k = 0; k < a->nseg; ++k) {
      const int nb = a->seg_b0[k + 1] - a->seg_b0[k];
      EMO_CHECK(nb > 0 && a->seg_T[k] > 0 && a->seg_row[k + 1] == a->seg_row[k] + (long)nb * a->seg_T[k] &&
                    (!a->pos || a->seg_prow[k + 1] == a->seg_prow[k] + 2L * a->seg_T[k] - 1),
                "attention: bad segment %d", k);
      tmax = std::max(tmax, a->seg_T[k]);
    }
    EMO_CHECK(tmax == a->Tq, "attention: Tq must be the longest segment (%d != %d)", a->Tq, tmax);
  }
  return 0;
}

// TK: the kernels' element type -- T, or f32s for T = float under dtype EMO_F32X3 (same memory, split-bf16 products)
template <typename T, typename TK = T>
int launch_fwd(const emoasr_attn_t& a_in, hipStream_t s) {
  emoasr_attn_t a = a_in;
  fill_seg_order(a);
  constexpr int LD = AttnCfg<TK>::LD;
  // waves per workgroup: the waves of this kernel never meet (wave-private LDS), so a workgroup is only a unit of dispatch; stacked
  // launches (thousands of query tiles of uneven length) are handed out wave by wave, which packs the CUs' wave slots tightest
  // key split (ks = 4): launches of at most two query tiles per CU -- decoding a few utterances -- where the tiles' own latency
  // (one key tile after the other) is the whole time; not with stored scores (the materialised backward) 
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
  }
  // training launches (bf16, relative positions, plain masks): the block-staged kernel, four query tiles per workgroup
  if constexpr (std::is_same<TK, bf16>::value) {
    constexpr int FW = 4;
    const long ntile = (long)cdiv(a.Tq, 32) * a.H * a.B;
    if (g_fwd4 && a.pos && !a.causal && !a.st && a.Tq == a.Tk && ((a.Tq >= 64 && ntile > 4L * n_cu) || g_fwd4 == 2)) {   // (option value 2: whenever eligible -- tests)
      dim3 g4(cdiv(a.Tq, 32 * FW), a.H, a.B);
      const int nt4 = (g_attn_xcd && a.nseg > 1) ? (int)g4.x : 0;
      if (nt4) g4 = dim3(8 * cdiv(a.H * a.B, 8) * nt4, 1, 1);
      emo_timer_begin(EMO_TIMER_ATTN_FWD, s);
#define EMO_FWD4_LAUNCH(TR_, DM_)                                                                  \
  do {                                                                                             \
    if (set_smem(attn_fwd4_kernel<TR_, FW, DM_>, Fwd4Cfg<FW>::smem())) return 1;                    \
    attn_fwd4_kernel<TR_, FW, DM_><<<g4, 64 * FW, Fwd4Cfg<FW>::smem(), s>>>(a, nt4);                \
  } while (0)
      const int dm = a.drop_p > 0.f ? (a.keep_mask ? 1 : 2) : 0;   // no dropout / keep-mask bits / inline hash
      if (g_tr) { if (dm == 1) EMO_FWD4_LAUNCH(true, 1); else if (dm == 2) EMO_FWD4_LAUNCH(true, 2); else EMO_FWD4_LAUNCH(true, 0); }
      else      { if (dm == 1) EMO_FWD4_LAUNCH(false, 1); else if (dm == 2) EMO_FWD4_LAUNCH(false, 2); else EMO_FWD4_LAUNCH(false, 0); }
#undef EMO_FWD4_LAUNCH
      emo_timer_end(EMO_TIMER_ATTN_FWD, s);
      EMO_LAUNCH_CHECK();
      return 0;
    }
  }
  const bool split = g_fwd_split && a.nseg <= 1 && !a.st && (long)cdiv(a.Tq, 32) * a.H * a.B <= 2L * n_cu;
  const int ks = split ? 4 : 1;
  const int nw = split ? 4 : (g_fwd_waves ? g_fwd_waves : (a.nseg > 1 ? 1 : 4));
  const int smem = nw * fwd_wave_bytes<TK>();
  dim3 grid(cdiv(a.Tq, split ? 32 : 32 * nw), a.H, a.B);   // (stacked micro-batches: Tq = the longest segment, B = all utterances)
  // stacked launches: 1-D, the query tiles of one (head, utterance) on one XCD (attn_block)
  const int nt = (g_attn_xcd && a.nseg > 1) ? (int)grid.x : 0;
  if (nt) grid = dim3(8 * cdiv(a.H * a.B, 8) * nt, 1, 1);
  // at most one wave per SIMD: see attn_fwd_kernel (not for the split type: two prefetch sets of (hi, lo) fragments spill)
  const bool one_round = !kSplit<TK> && (long)grid.x * grid.y * grid.z * nw <= 4L * n_cu;
  emo_timer_begin(EMO_TIMER_ATTN_FWD, s);
  if (g_tr) {
    if (one_round) {
      if (set_smem(attn_fwd_kernel<TK, true, true>, smem)) return 1;
      attn_fwd_kernel<TK, true, true><<<grid, 64 * nw, smem, s>>>(a, nt, ks);
    } else {
      if (set_smem(attn_fwd_kernel<TK, true, false>, smem)) return 1;
      attn_fwd_kernel<TK, true, false><<<grid, 64 * nw, smem, s>>>(a, nt, ks);
    }
  } else {
    if (one_round) {
      if (set_smem(attn_fwd_kernel<TK, false, true>, smem)) return 1;
      attn_fwd_kernel<TK, false, true><<<grid, 64 * nw, smem, s>>>(a, nt, ks);
    } else {
      if (set_smem(attn_fwd_kernel<TK, false, false>, smem)) return 1;
      attn_fwd_kernel<TK, false, false><<<grid, 64 * nw, smem, s>>>(a, nt, ks);
    }
  }
  emo_timer_end(EMO_TIMER_ATTN_FWD, s);
#ifdef EMO_ATTN_STAMP
  {  // debug builds: per-phase cycle counts of the first workgroup's wave 0, averaged over the key tiles (first call only)
    static int calls = 0;
    if (calls++ == 0 && sizeof(T) == 2) {
      unsigned long long h[64 * 8];
      hipStreamSynchronize(s);
      hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fwd_stamps), sizeof(h));
      const int nt = (a.Tk + 31) / 32 < 64 ? (a.Tk + 31) / 32 : 64;
      double acc[8] = {0};
      for (int t = 1; t < nt - 1; ++t)
        for (int k = 1; k < 7; ++k) acc[k] += (double)(h[t * 8 + k] - h[t * 8 + k - 1]);
      const char* nm[7] = {"", "fetch(PF2)", "V stash + S + band MFMA + Gs write", "fetch(!PF2)", "skew read + st", "softmax + dropout", "P.V"};
      fprintf(stderr, "[attn fwd stamp] B %d T %d one_round %d: per key tile (cycles):", a.B, a.Tq, (int)one_round);
      double tot = 0;
      for (int k = 1; k < 7; ++k) { fprintf(stderr, " %s %.0f |", nm[k], acc[k] / (nt - 2)); tot += acc[k] / (nt - 2); }
      fprintf(stderr, " total %.0f\n", tot);
    }
  }
#endif
  EMO_LAUNCH_CHECK();
  return 0;
}

template <typename T, bool TR, typename TK = T>
int launch_bwd_tr(const emoasr_attn_t& a, hipStream_t s, const int dtype) {   // dtype: the call's code (EMO_F32X3: the batched products below run split too)
  constexpr int LD = AttnCfg<T>::LD;
  const long rows = (long)a.B * a.Tq * a.H;
  attn_delta_kernel<T><<<cdiv(rows * 8, 256), 256, 0, s>>>(a);
  if (a.pdT && !a.st) {
    const int smem = 4 * DqCfg<TK>::WAVE_BYTES;
    dim3 grid(cdiv(a.Tq, 32), a.H, a.B);
    if (set_smem(attn_bwd_dq2_kernel<TK, TR>, smem)) return 1;
    attn_bwd_dq2_kernel<TK, TR><<<grid, 256, smem, s>>>(a);
    if (a.dbias_part && (a.dbias_u || a.dbias_v))
      attn_dbias_reduce_kernel<<<2 * a.H, 256, 0, s>>>(a.B * (int)grid.x, a.H, a.dbias_part, a.dbias_u,
                                                        a.pos ? a.dbias_v : nullptr);
  } else {
    const int smem = 4 * (DQ_GS_FLOATS * 4 + 96 * LD * (int)sizeof(T));
    dim3 grid(cdiv(a.Tq, 32), a.H, a.B);
    if (a.st && a.pdT) {
      if (set_smem(attn_bwd_dq_kernel<T, TR, true>, smem)) return 1;
      attn_bwd_dq_kernel<T, TR, true><<<grid, 256, smem, s>>>(a);
    } else {
      if (set_smem(attn_bwd_dq_kernel<T, TR, false>, smem)) return 1;
      attn_bwd_dq_kernel<T, TR, false><<<grid, 256, smem, s>>>(a);
    }
  }
  if (a.pdT) {
    // materialised mode: dV = Pd . dO, dK = dS . (Q + u), dpos_h = dBD_h^T . (Q_h + v)
    const long sp_b = (long)a.H * a.Tk * a.ldpd, sp_h = (long)a.Tk * a.ldpd;
    EMO_LAUNCH_CHECK();
    if (emoasr_gemm_nn_batched(dtype, a.Tk, DK, a.Tq, a.pdT, a.ldpd, sp_b, sp_h, a.dout, a.ldo,
                               (long)a.Tq * a.ldo, DK, a.dv, a.ldv, (long)a.Tk * a.ldv, DK, a.B, a.H, 1.f, 0, s))
      return 1;
    const bool dense_q = a.qu && a.qv && !a.st;  // Q+u / Q+v materialised by the delta pass
    const void* qk = dense_q ? a.qu : a.q;
    const long ldqk = dense_q ? (long)a.H * DK : a.ldq;
    if (emoasr_gemm_nn_batched(dtype, a.Tk, DK, a.Tq, a.dsT, a.ldpd, sp_b, sp_h, qk, ldqk,
                               (long)a.Tq * ldqk, DK, a.dk, a.ldk, (long)a.Tk * a.ldk, DK, a.B, a.H, 1.f, 0, s))
      return 1;
    if (a.bias_u && !dense_q) attn_dk_bias_kernel<T><<<cdiv((long)a.B * a.H * a.Tk, 4), 256, 0, s>>>(a);
    if (a.pos && a.dpos && dense_q) {
      const int R = 2 * a.Tq - 1;
      for (int h0 = 0; h0 < a.H; h0 += EMOASR_TN_GROUP_MAX) {
        emoasr_tn_problem_t pr[EMOASR_TN_GROUP_MAX];
        const int nh = min(a.H - h0, EMOASR_TN_GROUP_MAX);
        for (int i = 0; i < nh; ++i) {
          const int h = h0 + i;
          pr[i] = emoasr_tn_problem_t{R, DK, a.B * a.Tq,
                                      (const T*)a.dbd + (long)h * a.B * a.Tq * a.ldbd, a.ldbd,
                                      (const T*)a.qv + h * DK, ldqk,
                                      a.dpos + h * DK, (long)a.H * DK, 1.f, nullptr, 0.f};
        }
        if (emoasr_gemm_tn_grouped(dtype, nh, pr, s)) return 1;
      }
    } else if (a.pos && a.dpos) {
      const int R = 2 * a.Tq - 1;
      hipMemsetAsync(a.cs, 0, sizeof(float) * a.H * a.ldbd, s);
      // one product per head (reduction over all B*Tq query rows), grouped into one launch
      for (int h0 = 0; h0 < a.H; h0 += EMOASR_TN_GROUP_MAX) {
        emoasr_tn_problem_t pr[EMOASR_TN_GROUP_MAX];
        const int nh = min(a.H - h0, EMOASR_TN_GROUP_MAX);
        for (int i = 0; i < nh; ++i) {
          const int h = h0 + i;
          pr[i] = emoasr_tn_problem_t{R, DK, a.B * a.Tq,
                                      (const T*)a.dbd + (long)h * a.B * a.Tq * a.ldbd, a.ldbd,
                                      (const T*)a.q + h * DK, a.ldq,
                                      a.dpos + h * DK, (long)a.H * DK, 1.f,
                                      a.cs + (long)h * a.ldbd, 1.f};
        }
        if (emoasr_gemm_tn_grouped(dtype, nh, pr, s)) return 1;
      }
      attn_dpos_bias_kernel<<<cdiv((long)R * a.H * DK, 256), 256, 0, s>>>(a);
      if (a.st) {
        // dq += dBD . pos  (the (q+v) term), batched over (b, h); dbias_v from the column sums
        if (emoasr_gemm_nn_batched(dtype, a.Tq, DK, R, a.dbd, a.ldbd, (long)a.Tq * a.ldbd,
                                   (long)a.B * a.Tq * a.ldbd, a.pos, a.ldp, 0, DK, a.dq, a.ldq,
                                   (long)a.Tq * a.ldq, DK, a.B, a.H, 1.f, 1, s))
          return 1;
        if (a.dbias_v) attn_dbias_v_kernel<T><<<cdiv(a.H * DK, 256), 256, 0, s>>>(a);
      }
    }
    EMO_LAUNCH_CHECK();
    return 0;
  }
  {
    const int smem = 4 * (32 * 64 * 4 + 64 * LD * (int)sizeof(T));
    if (set_smem(attn_bwd_dkv_kernel<T, TR>, smem)) return 1;
    dim3 grid(cdiv(a.Tk, 128), a.H, a.B);
    attn_bwd_dkv_kernel<T, TR><<<grid, 256, smem, s>>>(a);
  }
  if (a.pos && a.dpos) {
    const int smem = 4 * (2 * 32 * 64 * 4 + 32 * LD * (int)sizeof(T));
    if (set_smem(attn_bwd_dpos_kernel<T, TR>, smem)) return 1;
    dim3 grid(cdiv(2 * a.Tq - 1, 32), a.H, a.B);
    attn_bwd_dpos_kernel<T, TR><<<grid, 256, smem, s>>>(a);
  }
  EMO_LAUNCH_CHECK();
  return 0;
}


struct FusedExtras { float* zero; long zero_n; const float* cast_src; void* cast_dst; long cast_n; };
FusedExtras g_fused_extras{};

// ---- side stream of the two-pass backward (option "attn_side", default on) ---------------------------------------------------------
// Two launches of an attention backward neither depend on nor feed its critical chain right away: the keep-mask bits (a pure
// function of seed and indices: they can be hashed while the layer's feed-forward / convolution backward runs) and the
// position-table gradient + its cast (consumed only by the layer's grouped weight-gradient launch at the very end).  Both are
// low-occupancy kernels (320 workgroups of the table gradient on 256 CUs), so on a second stream they run under the main chain.
//   emo_attn_bwd_prelaunch       prepare what the NEXT emoasr_attn_bwd_fused call with these arguments needs of FORWARD data only
//                                (keep mask, Q + bias copies, cleared table gradient) on the side stream
//   emo_attn_bwd_defer_join(1)   that call leaves the table gradient running on the side stream; emo_attn_bwd_join(stream) makes
//                                `stream` wait for it (csrc/layer.hip: right before the grouped weight-gradient launch)
// Callers that know nothing of this (ops.attn_bwd) get the join inside the call: same stream semantics as before.
int g_attn_side = 1;
int g_side_prio = 0;
hipStream_t g_side = nullptr;
hipEvent_t g_ev_fork = nullptr, g_ev_join = nullptr, g_ev_mask = nullptr;
bool g_join_pending = false, g_defer_join = false;
struct MaskTag { const void* buf = nullptr; uint64_t seed = 0; long nrows = -1; float p = 0.f; } g_mask_tag;

bool side_ready() {
  if (!g_attn_side) return false;
  if (g_side) return true;
  // option "attn_side_prio" = 1: the side stream at the LOWEST priority -- its launches (keep mask, Q + bias copies, table gradient)
  // are low-occupancy fillers that should take idle slots, not compete with the chain's kernels for dispatch
  int lo = 0, hi = 0;
  hipError_t ce = hipSuccess;
  if (g_side_prio && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess) ce = hipStreamCreateWithPriority(&g_side, hipStreamNonBlocking, lo);
  else ce = hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking);
  if (ce != hipSuccess) { g_side = nullptr; g_attn_side = 0; return false; }
  hipEventCreateWithFlags(&g_ev_fork, hipEventDisableTiming);
  hipEventCreateWithFlags(&g_ev_join, hipEventDisableTiming);
  hipEventCreateWithFlags(&g_ev_mask, hipEventDisableTiming);
  return true;
}
void side_join(hipStream_t s) {
  if (g_join_pending) { hipStreamWaitEvent(s, g_ev_join, 0); g_join_pending = false; }
}

struct FusedLayout { size_t dq32, qu, qv, dsq, mask, total; long ldds; int mask_nw; };
template <typename T>
FusedLayout fused_layout(const emoasr_attn_t& a, bool with_mask) {
  const bool rel = a.pos != nullptr;
  const long nrows = a.nseg > 1 ? a.seg_row[a.nseg] : (long)a.B * a.Tq;
  const long nqd = nrows * a.H * DK;
  FusedLayout L{};
  size_t off = 0;
  auto carve = [&](size_t n) { const size_t p = off; off += (n + 255) / 256 * 256; return p; };
  L.dq32 = carve((size_t)nqd * 4 * cdiv(a.Tk, 64));  // room for 64-key blocks (the finer grid of the single-pass kernel)
  if (rel) {
    L.qu = carve(nqd * sizeof(T));
    L.qv = carve(nqd * sizeof(T));
    L.ldds = (a.Tk + 31) / 32 * 32;
    L.dsq = carve((size_t)nrows * a.H * L.ldds * sizeof(T));
  }
  if (with_mask) {
    L.mask_nw = cdiv(a.Tk, 32);
    L.mask = carve((size_t)nrows * a.H * L.mask_nw * 4);
  }
  L.total = off;
  return L;
}

template <typename T>
int launch_dropmask(const emoasr_attn_t& a, unsigned* maskbuf, int nw, long nrows, hipStream_t s) {
  const int bx = (nw + 7) / 8 * 8;                               // words, padded to a multiple of 8 lanes
  EMO_CHECK(bx * a.H <= 1024, "attn_bwd_fused: H * ceil(Tk / 32) = %d exceeds one workgroup of the mask kernel", bx * a.H);
  const int rpb = std::max(1, 256 / (bx * a.H));                 // rows per (about) 256-thread block
  attn_dropmask_kernel<T><<<cdiv(nrows, rpb), dim3(bx, a.H, rpb), 0, s>>>(a, maskbuf, nw, nrows);
  return 0;
}

// the position-table gradient of one attention backward (attn_bwd_dpos3_kernel): one workgroup per (segment, head, tile diagonal,
// batch chunk); the chunks (<= 4) keep the grid a few rounds deep while a workgroup's one 16 KB flush stays amortised
template <typename T>
void launch_dpos3(const emoasr_attn_t& a, const FusedWs& ws, hipStream_t s) {
  Dpos3Plan pl{};
  const int nseg = a.nseg > 1 ? a.nseg : 1;
  int bmin = a.B;
  for (int k = 0; k < a.nseg && a.nseg > 1; ++k) bmin = std::min(bmin, a.seg_b0[k + 1] - a.seg_b0[k]);
  pl.nseg = nseg;
  pl.nch = std::max(1, std::min(4, bmin));
  for (int k = 0; k < nseg; ++k) {
    const int T_ = a.nseg > 1 ? a.seg_T[k] : a.Tq, nt = cdiv(T_, 32);
    pl.wg0[k + 1] = pl.wg0[k] + a.H * (2 * nt - 1) * pl.nch;
  }
  if (g_tr) attn_bwd_dpos3_kernel<T, true><<<pl.wg0[nseg], 256, 0, s>>>(a, ws, pl);
  else attn_bwd_dpos3_kernel<T, false><<<pl.wg0[nseg], 256, 0, s>>>(a, ws, pl);
}

template <typename T>
int launch_bwd_fused(const emoasr_attn_t& a_in, char* mem, size_t bytes, hipStream_t s) {
  emoasr_attn_t a = a_in;
  fill_seg_order(a);
  const bool rel = a.pos != nullptr;
  const long nrows = a.nseg > 1 ? a.seg_row[a.nseg] : (long)a.B * a.Tq;   // stacked micro-batches: all segments' rows
  const long nqd = nrows * a.H * DK;
  side_join(s);   // (a table gradient left on the side stream by a caller that never joined: it used this workspace)
  FusedWs ws{};
  const bool want_mask = g_bwd_split && a.drop_p > 0.f && !a.keep_mask;
  const FusedLayout lay = fused_layout<T>(a, want_mask);
  EMO_CHECK(lay.total <= bytes, "attn_bwd_fused: workspace too small (%zu < %zu bytes)", bytes, lay.total);
  ws.dq_slab = nqd;
  ws.dq32 = reinterpret_cast<float*>(mem + lay.dq32);
  T *qu = nullptr, *qv = nullptr;
  if (rel) {
    qu = reinterpret_cast<T*>(mem + lay.qu);
    qv = reinterpret_cast<T*>(mem + lay.qv);
    ws.ldds = lay.ldds;
    ws.dsq = mem + lay.dsq;
    ws.qu = qu; ws.qv = qv; ws.ldqu = (long)a.H * DK;
  } else {
    ws.qu = a.q; ws.qv = a.q; ws.ldqu = a.ldq;
  }
  unsigned* maskbuf = nullptr;
  const bool ext_mask = a.keep_mask != nullptr && a.drop_p > 0.f && g_bwd_split;   // (the single-pass kernel hashes inline: same mask)   // the forward's bits (emoasr_attn_dropmask): nothing to hash here
  if (ext_mask) {
    ws.mask_nw = a.keep_nw;
    ws.mask = a.keep_mask;
  } else if (want_mask) {
    ws.mask_nw = lay.mask_nw;
    maskbuf = reinterpret_cast<unsigned*>(mem + lay.mask);
    ws.mask = maskbuf;
  }
  const long rows = nrows * a.H;
  const FusedExtras fx = g_fused_extras;
  g_fused_extras = FusedExtras{};
#ifdef EMO_ATTN_STAMP
  static unsigned long long* d_stamp = nullptr;
  static int stamp_calls = 0;
  if (!d_stamp) hipMalloc(&d_stamp, 2 * 64 * 13 * 8);
  ws.stamp = stamp_calls++ == 0 ? d_stamp : nullptr;   // the first (eager) call only: later calls may be under stream capture
#endif
  // prepared ahead of time on the side stream (emo_attn_bwd_prelaunch)?  Then this prologue computes delta only.
  const bool prepared = g_bwd_split && g_mask_tag.buf == mem && g_mask_tag.seed == a.seed && g_mask_tag.nrows == nrows &&
                        g_mask_tag.p == a.drop_p;
  if (prepared) {
    hipStreamWaitEvent(s, g_ev_mask, 0);
    attn_bwd_prep_kernel<T><<<cdiv(rows * 8, 256), 256, 0, s>>>(a, nullptr, nullptr, nullptr, 0, nrows);
  } else {
    attn_bwd_prep_kernel<T><<<cdiv(rows * 8, 256), 256, 0, s>>>(a, qu, qv, fx.zero, fx.zero_n, nrows);
  }
  g_mask_tag = MaskTag{};
  // 4 key tiles per workgroup (one workgroup per CU) unless that grid spills into a second round of workgroups and the
  // 2-tile grid (two workgroups per CU) does not
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
  }
  const long nb4 = (long)cdiv(a.Tk, 128) * a.H * a.B, nb2 = (long)cdiv(a.Tk, 64) * a.H * a.B;
  const int fw = g_fused_fw ? g_fused_fw : ((nb4 > n_cu && nb2 <= 2 * n_cu) ? 2 : 4);
  if (g_bwd_split) {
    // two passes, two workgroups per CU each (see attn_bwd_kv_kernel); no dQ partial slabs, no finalize launch
    constexpr int FW = 4;
    using SC = SplitCfg<T, FW>;
    dim3 gk(cdiv(a.Tk, 32 * FW), a.H, a.B), gq(cdiv(a.Tq, 32 * FW), a.H, a.B);
    const int ntk = (g_attn_xcd && a.nseg > 1) ? (int)gk.x : 0, ntq = (g_attn_xcd && a.nseg > 1) ? (int)gq.x : 0;
    if (ntk) gk = dim3(8 * cdiv(a.H * a.B, 8) * ntk, 1, 1);
    if (ntq) gq = dim3(8 * cdiv(a.H * a.B, 8) * ntq, 1, 1);
    emo_timer_begin(EMO_TIMER_ATTN_BWD_MAIN, s);
    if (maskbuf && !prepared && launch_dropmask<T>(a, maskbuf, ws.mask_nw, nrows, s)) return 1;
#define EMO_KV_LAUNCH(TR_, REL_)                                                                                    \
  do {                                                                                                              \
    if (set_smem(attn_bwd_kv_kernel<T, TR_, REL_, FW>, SC::kv_smem(REL_))) return 1;                                \
    attn_bwd_kv_kernel<T, TR_, REL_, FW><<<gk, 64 * FW, SC::kv_smem(REL_), s>>>(a, ws, ntk);                        \
    if (ws.stamp) ws.stamp += 64 * 13;                                                                              \
  } while (0)
#define EMO_Q_LAUNCH(TR_, REL_)                                                                                     \
  do {                                                                                                              \
    if (REL_ && use_q2) {                                                                                           \
      if (set_smem(attn_bwd_q2_kernel<TR_, FW>, Q2Cfg<FW>::smem())) return 1;                                       \
      attn_bwd_q2_kernel<TR_, FW><<<gq, 64 * FW, Q2Cfg<FW>::smem(), s>>>(a, ws, ntq);                               \
    } else {                                                                                                        \
      if (set_smem(attn_bwd_q_kernel<T, TR_, REL_, FW>, SC::q_smem(REL_))) return 1;                                \
      attn_bwd_q_kernel<T, TR_, REL_, FW><<<gq, 64 * FW, SC::q_smem(REL_), s>>>(a, ws, ntq);                        \
    }                                                                                                               \
  } while (0)
    const bool use_q2 = g_q2 && rel && sizeof(T) == 2;
    ws.img_from_kv = use_q2 ? 1 : 0;
    if (rel) { if (g_tr) EMO_KV_LAUNCH(true, true); else EMO_KV_LAUNCH(false, true); }
    else     { if (g_tr) EMO_KV_LAUNCH(true, false); else EMO_KV_LAUNCH(false, false); }
    // the position-table gradient (+ its cast for the weight-gradient product): on the side stream when there is one
    hipStream_t st2 = s;
    const bool forked = rel && a.dpos && side_ready();
    auto table_grad = [&]() {
      if (forked) {
        hipEventRecord(g_ev_fork, s);
        hipStreamWaitEvent(g_side, g_ev_fork, 0);
        st2 = g_side;
      }
      if (rel && a.dpos) {
        emo_timer_begin(EMO_TIMER_ATTN_BWD_DPOS, st2);
        launch_dpos3<T>(a, ws, st2);
        emo_timer_end(EMO_TIMER_ATTN_BWD_DPOS, st2);
      }
    };
    // (forked right behind the KEY pass -- which writes the dS image since R6 -- the table gradient ran beside the query pass and
    // both stretched by what was hidden: query pass 2.93 -> 3.51 ms per step, the step 26.19 against 26.27 ms; it stays behind both)
    if (rel) { if (g_tr) EMO_Q_LAUNCH(true, true); else EMO_Q_LAUNCH(false, true); }
    else     { if (g_tr) EMO_Q_LAUNCH(true, false); else EMO_Q_LAUNCH(false, false); }
#undef EMO_KV_LAUNCH
#undef EMO_Q_LAUNCH
    emo_timer_end(EMO_TIMER_ATTN_BWD_MAIN, s);
    table_grad();
#ifdef EMO_ATTN_STAMP
    {  // debug builds: per-phase cycle counts of wave 0 of workgroup 0 of both passes, averaged over the sweep
      static int printed = 0;
      if (printed < 1 && ws.stamp) {
        ++printed;
        static unsigned long long h[2 * 64 * 13];
        hipStreamSynchronize(s);
        hipMemcpy(h, ws.stamp - 64 * 13, sizeof(h), hipMemcpyDeviceToHost);
        const int ns = (a.Tq + 31) / 32 < 64 ? (a.Tq + 31) / 32 : 64;
        const char* nm[3][10] = {{"", "loads+S+dP", "band+skew write", "skew read", "soft-max", "dV+dK", "barrier1", "stash", "fetch", "barrier2"},
                                 {"", "band+skew write+S+dP", "skew read", "soft-max+image", "dS store", "dQ", "barrier1", "stash", "fetch", "barrier2"},
                                 {"", "dS registers -> dG image", "dQ from K", "dQ from the band", "-", "-", "barrier1", "stash", "fetch", "barrier2"}};
        for (int kq = 0; kq < 2; ++kq) {
          double acc[10] = {0};
          int cnt = 0;
          for (int st = 1; st + 1 < ns; ++st, ++cnt)
            for (int k = 1; k < 10; ++k) acc[k] += (double)(h[kq * 64 * 13 + st * 13 + k] - h[kq * 64 * 13 + st * 13 + k - 1]);
          fprintf(stderr, "[attn stamp %s] B %d T %d: per step (cycles):", kq ? (use_q2 ? "q2" : "q") : "kv", a.B, a.Tq);
          double tot = 0;
          const int names = kq ? (use_q2 ? 2 : 1) : 0;
          for (int k = 1; k < 10; ++k) { fprintf(stderr, " %s %.0f", nm[names][k], acc[k] / (cnt > 0 ? cnt : 1)); tot += acc[k] / (cnt > 0 ? cnt : 1); }
          fprintf(stderr, " | total %.0f\n", tot);
          if (!kq || use_q2) {   // the key pass and the image-reading query pass also stamp their entry, loop start, loop end and exit
            const unsigned long long* w = h + kq * 64 * 13 + 60 * 13;
            fprintf(stderr, "[attn stamp %s] prologue %.0f  loop %.0f  epilogue %.0f cycles%s\n", kq ? "q2" : "kv", (double)(w[1] - w[0]),
                    (double)(w[2] - w[1]), (double)(w[3] - w[2]), "");
          }
        }
      }
    }
#endif
    if (fx.cast_n > 0)   // the finished f32 position-table gradient in the compute dtype, for its weight-gradient product
      attn_cast_kernel<T><<<(int)std::min<long>(cdiv(fx.cast_n, 256), 1024), 256, 0, st2>>>(fx.cast_src, (T*)fx.cast_dst, fx.cast_n);
    if (forked) {
      hipEventRecord(g_ev_join, g_side);
      g_join_pending = true;
      if (!g_defer_join) side_join(s);
    }
    g_defer_join = false;
    EMO_LAUNCH_CHECK();
    return 0;
  }
#define EMO_FUSED_LAUNCH(REL_, FW_)                                                                \
  do {                                                                                             \
    const int smem = FusedCfg<T, FW_>::smem_bytes(REL_);                                           \
    dim3 grid(cdiv(a.Tk, 32 * FW_), a.H, a.B);                                                     \
    const int nt = (g_attn_xcd && a.nseg > 1) ? (int)grid.x : 0;   /* key blocks of one (head, utterance) on one XCD */ \
    if (nt) grid = dim3(8 * cdiv(a.H * a.B, 8) * nt, 1, 1);                                        \
    if (g_tr) { if (set_smem(attn_bwd_fused_kernel<T, true, REL_, FW_>, smem)) return 1;           \
                attn_bwd_fused_kernel<T, true, REL_, FW_><<<grid, 64 * FW_, smem, s>>>(a, ws, nt); }   \
    else      { if (set_smem(attn_bwd_fused_kernel<T, false, REL_, FW_>, smem)) return 1;          \
                attn_bwd_fused_kernel<T, false, REL_, FW_><<<grid, 64 * FW_, smem, s>>>(a, ws, nt); }  \
  } while (0)
  emo_timer_begin(EMO_TIMER_ATTN_BWD_MAIN, s);
  if (rel) {
    if (fw == 2) EMO_FUSED_LAUNCH(true, 2); else EMO_FUSED_LAUNCH(true, 4);
    emo_timer_end(EMO_TIMER_ATTN_BWD_MAIN, s);
    if (a.dpos) {
      emo_timer_begin(EMO_TIMER_ATTN_BWD_DPOS, s);
      launch_dpos3<T>(a, ws, s);
      emo_timer_end(EMO_TIMER_ATTN_BWD_DPOS, s);
    }
  } else {
    if (fw == 2) EMO_FUSED_LAUNCH(false, 2); else EMO_FUSED_LAUNCH(false, 4);
    emo_timer_end(EMO_TIMER_ATTN_BWD_MAIN, s);
  }
#undef EMO_FUSED_LAUNCH
  FinSegs fsg{};
  if (a.nseg > 1) {
    fsg.n = a.nseg;
    for (int k = 0; k < a.nseg; ++k) { fsg.b0[k] = a.seg_b0[k]; fsg.T[k] = a.seg_T[k]; fsg.row[k] = a.seg_row[k]; }
    fsg.b0[a.nseg] = a.seg_b0[a.nseg]; fsg.row[a.nseg] = a.seg_row[a.nseg];
  }
  attn_bwd_fin_kernel<T><<<cdiv(nrows, 16), 256, 0, s>>>(nrows, a.H * DK, ws.dq32, ws.dq_slab, 32 * fw, a.Tq, a.Tk, a.klens,
                                                       (T*)a.dq, a.ldq, fx.cast_src, (T*)fx.cast_dst, fx.cast_n, fsg);
#ifdef EMO_ATTN_STAMP
  {  // debug builds: per-phase cycle counts of the main kernel's first workgroup (wave 0), averaged over the sweep
    static int printed = 0;
    if (printed < 1) {
      ++printed;
      unsigned long long h[64 * 13];
      hipStreamSynchronize(s);
      hipMemcpy(h, ws.stamp, sizeof(h), hipMemcpyDeviceToHost);
      const int ns = (a.Tq + 31) / 32 < 64 ? (a.Tq + 31) / 32 : 64;
      double acc[13] = {0};
      for (int st = 1; st < ns; ++st)
        for (int k = 1; k < 13; ++k) acc[k] += (double)(h[st * 13 + k] - h[st * 13 + k - 1]);
      fprintf(stderr, "[attn stamp] B %d T %d fw %d: per step (cycles):", a.B, a.Tq, fw);
      const char* nm[13] = {"", "S+band", "dP", "drop+exp+img", "dS store", "dQ", "dV+dK", "slab", "barrier1", "flush", "stash", "fetch", "barrier2"};
      double tot = 0;
      for (int k = 1; k < 13; ++k) { fprintf(stderr, " %s %.0f", nm[k], acc[k] / (ns - 1)); tot += acc[k] / (ns - 1); }
      fprintf(stderr, " | total %.0f\n", tot);
    }
  }
#endif
  EMO_LAUNCH_CHECK();
  return 0;
}
}  // namespace

// One-shot side jobs for the NEXT emoasr_attn_bwd_fused call of this thread's library state (csrc/layer.hip): clear `zero`
// (zn floats) in the prologue launch, and write cast_dst = compute-dtype(cast_src) (cn values) in the finalize launch.
void emo_attn_bwd_fused_extras(float* zero, long zn, const float* cast_src, void* cast_dst, long cn) {
  g_fused_extras = FusedExtras{zero, zn, cast_src, cast_dst, cn};
}

void emo_attn_set_side(int v) { g_attn_side = v ? 1 : 0; }
void emo_attn_set_side_prio(int v) { g_side_prio = v ? 1 : 0; }   // (takes effect when the side stream is created)
void emo_attn_bwd_defer_join(int v) { g_defer_join = v != 0; }
// With the keep mask handed over by the forward (round 6) the prelaunch has only the Q + bias copies left (~6 us of the call's own
// prologue), and a fork + a join of the side stream cost the main queue ~6 us EACH (the kernel sequence of the step shows the bubbles):
// 1 = prelaunch only when there is a mask to hash.  Option "attn_prelaunch" (value 1 = always prelaunch, as before).
int g_prelaunch_mask_only = 1;
void emo_attn_set_prelaunch(int v) { g_prelaunch_mask_only = v ? 0 : 1; }
void emo_attn_bwd_join(void* stream) { side_join((hipStream_t)stream); }
// Everything of the next emoasr_attn_bwd_fused(a, ws) call that depends on FORWARD data only, now, on the side stream: the keep
// mask (with dropout), the dense Q + pos_bias_u / Q + pos_bias_v copies, the clearing of the position-table gradient (zero, zn).
// No-op without a side stream; the call itself does whatever was not prepared.
int emo_attn_bwd_prelaunch(int dtype, const emoasr_attn_t* a_in, void* ws, size_t ws_bytes, float* zero, long zn, void* stream) {
  if (dtype != EMO_BF16 || !g_bwd_split || !side_ready()) return 0;
  emoasr_attn_t a = *a_in;
  fill_seg_order(a);
  const bool rel = a.pos != nullptr, want_mask = a.drop_p > 0.f && !a.keep_mask;
  if (!rel && !want_mask) return 0;
  if (!want_mask && g_prelaunch_mask_only) return 0;   // option "attn_prelaunch": see g_prelaunch_mask_only
  const FusedLayout lay = fused_layout<bf16>(a, want_mask);
  if (lay.total > ws_bytes) return 0;
  const long nrows = a.nseg > 1 ? a.seg_row[a.nseg] : (long)a.B * a.Tq;
  char* mem = static_cast<char*>(ws);
  // the regions' previous readers (the passes of the last call on this workspace) are ahead of this point on `stream`
  hipEventRecord(g_ev_fork, (hipStream_t)stream);
  hipStreamWaitEvent(g_side, g_ev_fork, 0);
  if (want_mask && launch_dropmask<bf16>(a, reinterpret_cast<unsigned*>(mem + lay.mask), lay.mask_nw, nrows, g_side)) return 1;
  if (rel) {
    emoasr_attn_t aq = a;
    aq.dout = nullptr;   // prologue kernel: the Q + bias copies and the clearing only (delta needs dO: the call's own prologue)
    attn_bwd_prep_kernel<bf16><<<cdiv(nrows * a.H * 8, 256), 256, 0, g_side>>>(aq, reinterpret_cast<bf16*>(mem + lay.qu),
                                                                              reinterpret_cast<bf16*>(mem + lay.qv), zero, zn, nrows);
  }
  hipEventRecord(g_ev_mask, g_side);
  g_mask_tag.buf = mem; g_mask_tag.seed = a.seed; g_mask_tag.nrows = nrows; g_mask_tag.p = a.drop_p;
  EMO_LAUNCH_CHECK();
  return 0;
}
void emo_attn_set_tr_read(int v) { g_tr = v; }
void emo_attn_set_lpt(int v) { g_attn_lpt = v; }
void emo_attn_set_fwd_split(int v) { g_fwd_split = v; }
void emo_attn_set_fwd4(int v) { g_fwd4 = v; }
void emo_attn_set_xcd(int v) { g_attn_xcd = v; }
void emo_attn_set_fwd_waves(int v) { g_fwd_waves = (v == 1 || v == 2 || v == 4) ? v : 0; }
void emo_attn_set_bwd_split(int v) { g_bwd_split = v ? 1 : 0; }
void emo_attn_set_q2(int v) { g_q2 = v ? 1 : 0; }
void emo_attn_set_fw(int v) { g_fused_fw = (v == 2 || v == 4) ? v : 0; }

extern "C" long emoasr_attn_dropmask_words(int Tk) { return cdiv(Tk, 32); }
// The keep mask of one attention launch as bits, for its forward AND its backward (emoasr_attn_t::keep_mask).  When the library has
// its side stream (option "attn_side") the hashing runs THERE, forked from `stream` at this point, and the first emoasr_attn_fwd
// given this mask waits for it: called at the top of a layer (csrc/layer.hip) the 50-60 us of integer hashing run under the
// macaron feed-forward block's products.
namespace {
bool g_fmask_pending = false;   // a mask is being hashed on the side stream: the next forward given a mask waits for g_ev_fmask
hipEvent_t g_ev_fmask = nullptr;
}
extern "C" int emoasr_attn_dropmask(int dtype, const emoasr_attn_t* a_in, unsigned* mask, int nw, void* stream) {
  EMO_CHECK(a_in && mask, "attn_dropmask: missing arguments");
  EMO_CHECK(nw >= cdiv(a_in->Tk, 32), "attn_dropmask: %d words per row for %d keys", nw, a_in->Tk);
  if (a_in->drop_p <= 0.f || a_in->B == 0 || a_in->Tq == 0) return 0;
  emoasr_attn_t a = *a_in;
  fill_seg_order(a);
  const long nrows = a.nseg > 1 ? a.seg_row[a.nseg] : (long)a.B * a.Tq;
  hipStream_t s = (hipStream_t)stream, st = s;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  hipStreamIsCapturing(s, &cap);
  const bool side = cap == hipStreamCaptureStatusNone && side_ready();
  if (side) {
    if (!g_ev_fmask) hipEventCreateWithFlags(&g_ev_fmask, hipEventDisableTiming);
    hipEventRecord(g_ev_fork, s);
    hipStreamWaitEvent(g_side, g_ev_fork, 0);
    st = g_side;
  }
  int rc;
  if (dtype == EMO_BF16) rc = launch_dropmask<bf16>(a, mask, nw, nrows, st);
  else rc = launch_dropmask<float>(a, mask, nw, nrows, st);
  if (rc) return 1;
  if (side) {
    hipEventRecord(g_ev_fmask, g_side);
    g_fmask_pending = true;
  }
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_attn_fwd(int dtype, const emoasr_attn_t* a, void* stream) {
  if (check_args(a, dtype)) return 1;
  if (a->B == 0 || a->Tq == 0) return 0;
  EMO_CHECK(!a->keep_mask || a->keep_nw >= cdiv(a->Tk, 32), "attn_fwd: keep mask of %d words per row for %d keys", a->keep_nw, a->Tk);
  if (a->keep_mask && g_fmask_pending) {   // hashed on the side stream (emoasr_attn_dropmask): join; later calls on this stream are behind the wait
    hipStreamWaitEvent((hipStream_t)stream, g_ev_fmask, 0);
    g_fmask_pending = false;
  }
  if (dtype == EMO_F32X3) return launch_fwd<float, f32s>(*a, (hipStream_t)stream);
  EMO_DISPATCH(dtype, return (launch_fwd<T>(*a, (hipStream_t)stream)));
  return 0;
}

extern "C" int emoasr_attn_bwd(int dtype, const emoasr_attn_t* a, void* stream) {
  if (check_args(a, dtype)) return 1;
  EMO_CHECK(a->dout && a->out && a->delta && a->dq && a->dk && a->dv, "attn_bwd: missing buffers");
  EMO_CHECK(a->nseg <= 1, "attn_bwd: stacked segments run emoasr_attn_bwd_fused");
  if (a->pdT) {
    EMO_CHECK(a->dsT && a->ldpd >= a->Tq && a->ldpd % 8 == 0, "attn_bwd: bad pdT/dsT scratch");
    EMO_CHECK(!a->pos || !a->dpos || (a->dbd && a->cs && a->ldbd >= 2 * a->Tq - 1 && a->ldbd % 8 == 0),
              "attn_bwd: bad dbd/cs scratch");
    EMO_CHECK(!a->st || a->ldst >= a->Tq, "attn_bwd: bad st");
  }
  if (a->B == 0 || a->Tq == 0) return 0;
  // dtype EMO_F32X3: the score-recomputing dQ kernel (the training path's: pdT without stored scores) runs its
  // products split; the GEMMs behind it (dV, dK, dpos) follow the same option inside gemm.hip
  if (dtype == EMO_F32X3 && a->pdT && !a->st) return launch_bwd_tr<float, true, f32s>(*a, (hipStream_t)stream, dtype);
  EMO_DISPATCH(dtype, {
    if (g_tr) return (launch_bwd_tr<T, true>(*a, (hipStream_t)stream, dtype));
    return (launch_bwd_tr<T, false>(*a, (hipStream_t)stream, dtype));
  });
  return 0;
}

// rows = B * Tq of a dense batch, or all rows of stacked micro-batches whose longest segment has Tk frames
// The materialised backward's scratch as two caller-provided areas (what emoasr_amd/ops.py: AttnScratch allocates tensor by
// tensor): `images` -- P^T, dS^T and the dBD band, which must be ZERO before the first call for a given (B, Tq, Tk, klens) and may
// then be reused by calls with the same masks -- and `scratch` (cs, Q+u, Q+v, the dbias partials: no initialisation).
namespace {
struct MatLayout { size_t pdT, dsT, dbd, img_bytes, cs, qu, qv, part, scr_bytes; long ldpd, ldbd; };
MatLayout mat_layout(int dtype, int B, int H, int Tq, int Tk, int rel) {
  const size_t esz = dtype == EMO_BF16 ? 2 : 4;
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  MatLayout m{};
  m.ldpd = (Tq + 7) / 8 * 8;
  m.ldbd = (2L * Tq - 1 + 7) / 8 * 8;
  size_t o = 0;
  m.pdT = o; o += up((size_t)B * H * Tk * m.ldpd * esz);
  m.dsT = o; o += up((size_t)B * H * Tk * m.ldpd * esz);
  m.dbd = o; if (rel) o += up((size_t)H * B * Tq * m.ldbd * esz);
  m.img_bytes = o;
  o = 0;
  m.cs = o; if (rel) o += up((size_t)H * m.ldbd * 4);
  m.qu = o; if (rel) o += up((size_t)B * Tq * H * DK * esz);
  m.qv = o; if (rel) o += up((size_t)B * Tq * H * DK * esz);
  m.part = o; o += up((size_t)B * cdiv(Tq, 32) * H * 2 * DK * 4);
  m.scr_bytes = o;
  return m;
}
}  // namespace

extern "C" size_t emoasr_attn_bwd_mat_bytes(int dtype, int B, int H, int Tq, int Tk, int rel, int which) {
  if (B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0) return 0;
  const MatLayout m = mat_layout(dtype, B, H, Tq, Tk, rel);
  return which == 0 ? m.img_bytes : m.scr_bytes;
}

extern "C" int emoasr_attn_bwd_mat_bind(int dtype, emoasr_attn_t* a, void* images, void* scratch) {
  EMO_CHECK(a && images && scratch, "attn_bwd_mat_bind: missing arguments");
  EMO_CHECK(a->DK == DK, "attn_bwd_mat_bind: head dimension %d unsupported", a->DK);
  const int rel = a->pos != nullptr;
  const MatLayout m = mat_layout(dtype, a->B, a->H, a->Tq, a->Tk, rel);
  char *im = static_cast<char*>(images), *sc = static_cast<char*>(scratch);
  a->pdT = im + m.pdT; a->dsT = im + m.dsT; a->ldpd = m.ldpd;
  a->dbd = rel ? im + m.dbd : nullptr; a->ldbd = rel ? m.ldbd : 0;
  a->cs = rel ? reinterpret_cast<float*>(sc + m.cs) : nullptr;
  const bool dense = rel && a->bias_u && a->bias_v;
  a->qu = dense ? sc + m.qu : nullptr; a->qv = dense ? sc + m.qv : nullptr;
  a->dbias_part = reinterpret_cast<float*>(sc + m.part);
  return 0;
}

extern "C" size_t emoasr_attn_bwd_fused_ws_bytes_rows(int dtype, long rows, int H, int Tk, int rel) {
  const size_t esz = dtype == EMO_BF16 ? 2 : 4;
  auto up = [](size_t n) { return (n + 255) / 256 * 256; };
  const size_t nqd = (size_t)rows * H * DK;
  size_t n = up(nqd * 4 * ((Tk + 63) / 64));
  if (rel) n += 2 * up(nqd * esz) + up((size_t)rows * H * ((Tk + 31) / 32 * 32) * esz);
  n += up((size_t)rows * H * ((Tk + 31) / 32) * 4);   // keep-mask bits of the two-pass backward
  return n;
}

extern "C" size_t emoasr_attn_bwd_fused_ws_bytes(int dtype, int B, int H, int Tq, int Tk, int rel) {
  return emoasr_attn_bwd_fused_ws_bytes_rows(dtype, (long)B * Tq, H, Tk, rel);
}

extern "C" int emoasr_attn_bwd_fused(int dtype, const emoasr_attn_t* a, void* ws, size_t ws_bytes, void* stream) {
  if (check_args(a, dtype)) return 1;
  EMO_CHECK(dtype == EMO_BF16, "attn_bwd_fused: bf16 only (f32 runs emoasr_attn_bwd)");
  EMO_CHECK(!a->causal, "attn_bwd_fused: causal masks run emoasr_attn_bwd");
  EMO_CHECK(a->dout && a->out && a->delta && a->dq && a->dk && a->dv && ws, "attn_bwd_fused: missing buffers");
  EMO_CHECK(!a->pos || (a->bias_u && a->bias_v), "attn_bwd_fused: relative positions need both position biases");
  EMO_CHECK((a->dbias_u != nullptr) == (a->dbias_v != nullptr), "attn_bwd_fused: dbias_u and dbias_v go together");
  EMO_CHECK(a->H <= 32, "attn_bwd_fused: H=%d unsupported (at most 32 heads)", a->H);
  if (a->B == 0 || a->Tq == 0) return 0;
  return launch_bwd_fused<bf16>(*a, static_cast<char*>(ws), ws_bytes, (hipStream_t)stream);
}
