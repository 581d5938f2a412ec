// Persistent, weight-stationary bf16 GEMM for the K = 256 products of the Conformer layers over MANY rows:
//
//   C[M,N] = epilogue(A[M,256] . B[N,256]^T)        N a multiple of 256, M >= a few thousand rows (stacked micro-batches)
//
// feed-forward w1 (N 1024), q/k/v (768), pointwise conv 1 (512), attention output / pointwise conv 2 (256) and the data
// gradients whose reduction runs over a 256-wide output (feed-forward w2, attention output, pointwise conv 2):
// asr/modeling/transformer.py:62-71,102-118, conformer.py:103-117 and their autograd.
//
// Why another kernel: with M ~ 35 k rows these products are 5-18 GFLOP each, and both tiled kernels (gemm.hip 128 x 64,
// gemm_big.hip 128 x 256) spend a tile's life serially -- stage the operands (one memory round trip), four k-tiles of MFMAs
// (~1.7 us), write the tile out -- with nothing overlapping across tiles: 4.3 rounds of ~8 us for the 35 k x 1024 product
// (profiles/r03_gemm_35k_rows.txt: 37 us standalone, 57-69 us with the training epilogue; the HBM floor is 14 us).
//
// Here a workgroup is PERSISTENT and the weight is STATIONARY:
//   * 320 threads = 4 consumer waves + 1 loader wave, one workgroup per CU.  A workgroup owns one 256-column slab of B for its
//     whole life: consumer wave w keeps the fragments of columns 64 w .. 64 w + 63 for all of K in REGISTERS (32 x bf16x8 =
//     128 VGPRs, loaded once).
//   * It walks the 64-row tiles of A assigned to it.  The loader wave moves A tiles by LDS-DMA (buffer_load ... lds, no staging
//     registers) into a ring of three 32 KB stages, TWO tiles ahead of the MFMAs; the image is four 64-deep k-tiles of
//     [64 rows][128 B] with the 16-byte chunks XOR-swizzled on the source side (conflict-free ds_read_b128 fragments, as
//     gemm_big.hip); its only waits are counted ones on its own DMA, one s_barrier per tile publishes a tile to the consumers.
//   * Per tile a consumer multiplies 64 rows x its 64 columns: 128 v_mfma_f32_16x16x32_bf16, each A fragment read once from LDS
//     and used against four column groups (LDS read rate: half the MFMA time), then runs the epilogue of that tile; its stores
//     drain in the background (nothing in a consumer ever waits for them), its residual / saved-activation operands are
//     requested before the MFMAs.
//   * Tiles are dealt so that the workgroups working on the SAME rows (one per slab) sit on the same XCD and share its L2.
//
// STATUS (round 3, MI355X, M = 35 145 rows; profiles/r03_gemm_k256.txt): NOT the default -- built only with EMOASR_EXPERIMENTAL=1.
//   product 35145 x 1024 x 256      tiled kernels   this kernel   this kernel without its epilogue
//     plain (bias)                      37.1 us        47.2 us        21.6 us  (DMA + MFMAs: the HBM floor is ~14 us)
//     w1 (bias, Swish, dropout, u)      68.7 us        79.5 us
//   The streaming part does what it was built for (the loader keeps two tiles ahead, the MFMA phase runs back to back), but
//   at ONE consumer wave per SIMD the epilogue's LDS round trips and VALU work (2.9 us per 64-row tile and wave) are exposed
//   in full, where the tiled kernels overlap them across 2-3 resident workgroups per CU.  Removing that needs a second set of
//   consumer waves per SIMD, which the 128 B-fragment registers per wave do not leave room for (256 registers per wave at two
//   waves per SIMD).  Lessons kept in the default path: the pairwise dropout hash of the GEMM epilogues (common.h:
//   dropout_apply8; the K = 256 products with a dropout epilogue lost 15-20 % of their time), loop-invariant loads pinned
//   before a loop that also stores, no LDS-DMA in a wave that stores.
//
// Epilogue: exactly emoasr_gemm_nt's (include/emoasr_hip.h): alpha, bias, pre_out, activation, dact, dropout (mask index
// row * N + col), residual -- same order, same roundings, so the results are bit-identical to the tiled kernels' as long as the
// k order of the accumulation (ascending in 32-deep slices here as there) matches to f32 rounding; tests compare the two.
#include <algorithm>
#include "../mma.h"
#include "../../../include/emoasr_hip.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_void_ptr_k;

struct K256Args {
  int M, N;
  const void* A; long lda;
  const void* B; long ldb;
  void* C; long ldc;
  emoasr_epilogue_t ep;
  int nslab;       // N / 256
  int ntile;       // ceil(M / 64)
  int streams;     // workgroups per slab (each walks tiles stream, stream + streams, ...)
  int dbg;         // timing ablations (option "gemm_k256_dbg"): 1 no epilogue, 2 no MFMAs, 4 no DMA, 8 no tile loop at all
};

constexpr int KT_ROWS = 64;                  // rows per A tile
constexpr int KT_STAGE = KT_ROWS * 512;      // bytes per stage: 4 k-tiles x 64 rows x 128 B
constexpr int KT_NSTAGE = 3;
constexpr int KT_FLD = 68;                   // floats per row of the wave-private epilogue slab
constexpr int KT_SLAB = 16 * KT_FLD * 4;     // bytes per wave
constexpr int KT_SMEM = KT_NSTAGE * KT_STAGE + 4 * KT_SLAB;

__device__ __forceinline__ void dma16k(__amdgpu_buffer_rsrc_t r, char* lds, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_ptr_k)lds, 16, voff, soff, 0, 0);
}

// 320 threads: waves 0..3 multiply and run the epilogue ("consumers"), wave 4 only moves A tiles into LDS ("loader").
// Why a loader wave: a wave that has LDS-DMA in flight AND issues stores cannot wait for one without the other (one in-order
// counter per wave, and the compiler must assume a pending LDS-DMA may alias any LDS access: it drained vmcnt to 0 in front of
// every epilogue slab access -- 5 us per tile in the first version of this kernel, profiles/r03_gemm_k256.txt).  With the roles
// split, the consumers never wait for memory except for their own epilogue operands, their stores drain in the background, and
// the loader's only waits are counted ones on its own DMA.
__global__ __launch_bounds__(320, 1) void gemm_k256_kernel(const K256Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- which slab, which stream of tiles: the `nslab` workgroups of one stream sit on the same XCD (blockIdx % 8) ----------
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int slab = loc % g.nslab;
  const int stream = (loc / g.nslab) * 8 + xcd;
  if (stream >= g.streams) return;
  int my_tiles = g.ntile > stream ? (g.ntile - stream + g.streams - 1) / g.streams : 0;
  if (g.dbg & 8) my_tiles = 0;
  if (my_tiles == 0) return;

  if (wave == 4) {
    // ================================ loader: A tiles -> LDS ring, two tiles ahead =================================================
    // a tile = 32 pieces of 1 KiB (8 rows x 128 B of one k-tile): piece p = k-tile p >> 3, rows 8 (p & 7) .. + 7; a lane fetches
    // row 8 (p & 7) + (lane >> 3), 16-byte chunk (lane & 7) ^ swizzle(row); the piece lands lane-linear at stage + 1024 p
    const int sub = lane >> 3, pc = lane & 7;
    unsigned d_off[32];
#pragma unroll
    for (int p = 0; p < 32; ++p) {
      const int kt = p >> 3, row = 8 * (p & 7) + sub;
      const unsigned ch = (unsigned)(pc ^ ((row >> 1) & 7)) * 16u;
      d_off[p] = (unsigned)((long)row * g.lda * 2) + (unsigned)kt * 128u + ch;
    }
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(g.A);
    const unsigned a_bytes_row = (unsigned)(g.lda * 2);
    auto issue_tile = [&](int tile, int stage) __attribute__((always_inline)) {
      const int m0 = tile * KT_ROWS;
      char* dst = smem + stage * KT_STAGE;
      if (g.dbg & 4) return;
#pragma unroll
      for (int p = 0; p < 32; ++p) {
        const int row = 8 * (p & 7) + sub;
        const unsigned v = (m0 + row < g.M) ? d_off[p] : EMO_OOB;   // rows past M: zeros land in LDS
        dma16k(rsA, dst + p * 1024, v, (unsigned)m0 * a_bytes_row);
      }
    };
    issue_tile(stream, 0);
    if (my_tiles > 1) issue_tile(stream + g.streams, 1);
    for (int it = 0; it < my_tiles; ++it) {
      // tile `it` has landed once at most the next tile's 32 DMA instructions are outstanding (loads complete in order)
      if (it + 1 < my_tiles) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // publishes tile `it`; the consumers have finished reading tile it - 1's stage
      if (it + 2 < my_tiles) issue_tile(stream + (it + 2) * g.streams, (it + 2) % KT_NSTAGE);
    }
    return;
  }

  // ==================================== consumers ====================================================================================
  const int n0 = slab * 256 + wave * 64;
  // ---- the wave's share of B, for all of K, in registers ---------------------------------------------------------------------
  // fragment (j, ks): columns n0 + 16 j + (lane & 15), k = 32 ks + 8 (lane >> 4) .. + 7
  bf16x8 bq[4][8];
  {
    const bf16* Bp = static_cast<const bf16*>(g.B);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bf16* rowp = Bp + (long)(n0 + 16 * j + (lane & 15)) * g.ldb + 8 * (lane >> 4);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) bq[j][ks] = *reinterpret_cast<const bf16x8*>(rowp + 32 * ks);
    }
  }
  // ---- fragment read offsets (gemm_big.hip's scheme): lane -> row (lane & 15) of a 16-row group, chunk swizzled ------------------
  const int frow = lane & 15;
  const unsigned fsw = (unsigned)((lane >> 1) & 7);
  unsigned fch[2];
  fch[0] = (((unsigned)(lane >> 4)) ^ fsw) * 16u;
  fch[1] = (((unsigned)(4 + (lane >> 4))) ^ fsw) * 16u;
  const unsigned a_frag0 = (unsigned)(frow * 128);

  const emoasr_epilogue_t& ep = g.ep;
  const bf16* __restrict__ res = static_cast<const bf16*>(ep.residual);
  const bf16* __restrict__ dpre = static_cast<const bf16*>(ep.dact_pre);
  bf16* __restrict__ pre_out = static_cast<bf16*>(ep.pre_out);
  bf16* __restrict__ Cp = static_cast<bf16*>(g.C);
  float* fslab = reinterpret_cast<float*>(smem + KT_NSTAGE * KT_STAGE) + wave * (16 * KT_FLD);
  // a lane's 8 output columns are the same for every row it handles: the bias is loaded once
  f32x4 bias0 = f32x4{0.f, 0.f, 0.f, 0.f}, bias1 = bias0;
  if (ep.bias) {
    bias0 = *reinterpret_cast<const f32x4*>(ep.bias + n0 + (lane & 7) * 8);
    bias1 = *reinterpret_cast<const f32x4*>(ep.bias + n0 + (lane & 7) * 8 + 4);
  }
  // A use of every loop-invariant load right here: otherwise the compiler's wait for them sits at their first use INSIDE the tile
  // loop -- as s_waitcnt vmcnt(0), since it cannot tell the first iteration from the others -- and there it also drains the
  // previous tile's stores in front of every tile's MFMAs and every chunk's bias add (3.3 us per tile, measured).
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(bq[j][ks]));
  asm volatile("" : "+v"(bias0), "+v"(bias1));

  // One operand of the epilogue at most is read back per element (residual in the forward products, the saved pre-activation in the
  // data gradients; the host routes launches with both to the tiled kernels): it is requested before the MFMAs of its rows.
  const bf16* __restrict__ extra = res ? res : dpre;
  const long ld_extra = res ? (long)ep.ldr : g.ldc;
  // A tile is processed as two halves of 32 rows (accumulators: 32 registers; with the 128 of B, the prefetched operand and the
  // epilogue's temporaries the consumer stays under the 256 registers two waves per SIMD -- consumer + loader -- leave it).
  // (the operand of half-step hs + 1 is requested BEFORE the stores of half-step hs: a wave's memory operations complete in order,
  // so a load issued behind those stores could only be consumed once they had all been acknowledged)
  auto prefetch = [&](bf16x8 (&dst)[2][2], const int hs) __attribute__((always_inline)) {
    if (!extra || hs >= 2 * my_tiles) return;
    const int m0p = (stream + (hs >> 1) * g.streams) * KT_ROWS + (hs & 1) * 32;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int id = lane + 64 * hh, row = id >> 3, cc = id & 7;
        const int grow = min(m0p + i * 16 + row, g.M - 1);   // (rows past M: any valid address; never stored)
        dst[i][hh] = *reinterpret_cast<const bf16x8*>(extra + (long)grow * ld_extra + n0 + cc * 8);
      }
  };
  bf16x8 pex[2][2], pnx[2][2];
  prefetch(pex, 0);
  for (int it = 0; it < my_tiles; ++it) {
    const int m0 = (stream + it * g.streams) * KT_ROWS;
    const char* sta = smem + (it % KT_NSTAGE) * KT_STAGE;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h == 0) __builtin_amdgcn_s_barrier();   // the loader has landed tile `it` (and every consumer is past tile it - 1)
      f32x4 acc[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      // 8 k-slices of 32: slice ks lives in k-tile ks >> 1, half ks & 1
      if (!(g.dbg & 2)) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          bf16x8 aq[2];
#pragma unroll
          for (int i = 0; i < 2; ++i)
            aq[i] = *reinterpret_cast<const bf16x8*>(sta + (ks >> 1) * 8192 + a_frag0 + (2 * h + i) * 2048 + fch[ks & 1]);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[j][ks], aq[i], acc[i][j], 0, 0, 0);
        }
      }
      prefetch(pnx, 2 * it + h + 1);
      if (g.dbg & 1) continue;
      // ---- epilogue: 16-row slabs through wave-private LDS (f32), then every lane owns 8 consecutive columns of a row -- full
      //      128-byte row segments for every global access; emoasr_gemm_nt's order of operations -----------------------------------
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(fslab + frow * KT_FLD + j * 16 + 4 * (lane >> 4)) = acc[i][j];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int id = lane + 64 * hh, row = id >> 3, cc = id & 7;
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(fslab + row * KT_FLD + cc * 8);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(fslab + row * KT_FLD + cc * 8 + 4);
          const int grow = m0 + (2 * h + i) * 16 + row;
          const int col = n0 + cc * 8;
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = v0[e]; v[4 + e] = v1[e]; }
          if (ep.bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = ep.alpha * v[e] + bias0[e]; v[4 + e] = ep.alpha * v[4 + e] + bias1[e]; }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= ep.alpha;
          }
          const long off = (long)grow * g.ldc + col;
          const bool ok = grow < g.M;
          if (pre_out && ok) store8<bf16>(pre_out + off, v);
          act_vec<8>(ep.act, v);
          if (dpre) {
            float d[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e] = (float)pex[i][hh][e];
            dact_vec<8>(ep.dact, d, v);
          }
          dropout_apply8(ep.seed, (uint64_t)grow * (uint64_t)g.N + col, ep.drop_p, v);   // (N % 256 == 0, col % 8 == 0)
          if (res) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)pex[i][hh][e] + ep.res_scale * v[e];
          }
          if (ok) store8<bf16>(Cp + off, v);
        }
      }
      if (extra) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) pex[i][hh] = pnx[i][hh];
      }
    }
  }
}

int g_k256_dbg = 0;
int g_gemm_k256 = 1;            // option "gemm_k256": 0 = these products stay on the tiled kernels (A/B switch)
int g_k256_min_rows = 8192;     // below this the persistent walk has too few tiles per workgroup to pay for loading the slab

int k256_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  }
  return n;
}

}  // namespace

void emo_gemm_set_k256(int v) { g_gemm_k256 = v; }
void emo_gemm_set_k256_dbg(int v) { g_k256_dbg = v; }
void emo_gemm_set_k256_min_rows(int v) { g_k256_min_rows = v > 0 ? v : 1; }

// Does the persistent K = 256 kernel take this emoasr_gemm_nt call?
bool emo_gemm_nt_k256_wants(int M, int N, int K, long lda, long ldb, long ldc, const emoasr_epilogue_t& ep) {
  if (!g_gemm_k256 || K != 256 || N % 256 != 0 || N < 256 || N > 2048 || M < g_k256_min_rows) return false;
  if (lda % 8 != 0 || ldb % 8 != 0 || ldc % 8 != 0 || ep.out_f32) return false;
  if (ep.residual && ep.ldr % 8 != 0) return false;
  if (ep.residual && ep.dact_pre) return false;   // (one read-back operand per element: see the kernel)
  if ((ep.act & EMO_ACT_SAVE_DACT) || ep.dact == EMO_DACT_MUL) return false;   // (the saved-factor epilogue is not built here)
  if ((long)M * lda * 2 >= (1L << 32)) return false;   // (the DMA descriptor's 32-bit offsets)
  return true;
}

int emo_gemm_nt_k256(int M, int N, const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                     const emoasr_epilogue_t& ep, hipStream_t s) {
  K256Args a{};
  a.M = M; a.N = N; a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc; a.ep = ep;
  a.nslab = N / 256;
  a.dbg = g_k256_dbg;
  a.ntile = cdiv(M, KT_ROWS);
  // one workgroup per CU: streams per slab = CUs / slabs, rounded down to whole groups of 8 XCD-mates where possible
  const int cus = k256_cus();
  int streams = std::max(1, cus / a.nslab);
  if (streams >= 8) streams -= streams % 8;   // whole groups of 8 XCD-mates: the grid then never exceeds one round of CUs
  streams = std::min(streams, a.ntile);
  a.streams = streams;
  // grid: index = 8 * (slab + nslab * (stream / 8)) + stream % 8  (see the kernel's decode); pad to cover every stream
  const int groups = cdiv(streams, 8);
  const int grid = 8 * a.nslab * groups;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_k256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, KT_SMEM);
    if (e != hipSuccess) { emo_set_error("hipFuncSetAttribute(%d): %s", KT_SMEM, hipGetErrorString(e)); return 1; }
    attr_done = true;
  }
  gemm_k256_kernel<<<grid, 320, KT_SMEM, s>>>(a);
  EMO_LAUNCH_CHECK();
  return 0;
}
