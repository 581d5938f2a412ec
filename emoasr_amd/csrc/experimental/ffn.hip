// Fused position-wise feed-forward block of the Conformer / Transformer layers (bf16, d = 256):
//
//     y = x + res_scale * drop( W2 . drop(act( W1 . LN(x) + b1 )) + b2 )
//
// one launch instead of LayerNorm + two GEMMs (asr/modeling/transformer.py:102-118, the macaron halves of
// conformer.py:196-207,222-225).  A workgroup owns 64 rows of x for the whole block:
//   * LayerNorm in registers (one wave per row at a time, the arithmetic of ln_fwd_kernel), h = LN(x) goes to HBM
//     (the weight-gradient product needs it) and, through LDS, into MFMA operand fragments that stay in registers;
//   * the F-wide intermediate is produced and consumed in chunks of 256 columns: U^T = W1c . h^T (transposed, so that
//     four consecutive intermediate columns of a row sit in consecutive accumulator registers), bias + Swish/ReLU +
//     counter-RNG dropout in registers, the pre-activation u and the activation a leave as bf16 LDS images that are (i)
//     copied out row-major with 16-byte stores (the backward needs both) and (ii) read back as the k-contiguous A
//     operand of y += a_c . W2c^T -- the intermediate never comes back from HBM;
//   * the weights stream through ONE flat sequence of [256 rows][64 k] tiles (W1 chunk, then the W2 chunk, ...) with
//     a register ring (three tiles in flight) and two LDS buffers: one barrier per tile, 16 MFMAs per wave per tile;
//   * epilogue: bias, dropout, residual exactly as the GEMM epilogue (gemm.hip: nt_epilogue) computes them.
// Every product accumulates in the same k order as the unfused kernels and every rounding happens at the same
// place, so the outputs (h, mean, rstd, u, a, y) are bit-identical to emoasr_layernorm_fwd + 2 x emoasr_gemm_nt
// (tests/test_ops_gpu.py::test_ffn_fwd_fused_is_bit_identical_to_layernorm_plus_two_gemms).
//
// MEASURED (round 2, MI355X, M = 7029, F = 1024): 87 us against 40 us for the three unfused launches, so the layer
// runtime does NOT use it by default (emoasr_set_option("ffn_fused", 1) switches it on).  Ablation (profiles/
// r02_ffn_fused.txt): with 64 rows per workgroup only 110 of the 256 CUs have work, each CU has to stream the full 1 MB of
// weights for its 64 rows, and at one wave per SIMD (468 registers) staging a 32 KB tile through registers costs ~2.3 k
// cycles (8 buffer loads + 8 ds_write_b128 per lane: ~1.1 us each way per tile) against 512 cycles of MFMA -- the unfused
// GEMMs run three workgroups per CU on all CUs and hide exactly that.  The kernel is kept as the tested reference point
// for a version with LDS-DMA staging and more rows per weight byte.
#include <type_traits>
#include "../mma.h"
#include "../../../include/emoasr_hip.h"

namespace {

constexpr int FD = 256;    // model dimension (row length of x, k extent of W1, rows of W2)
constexpr int FBM = 64;    // rows per workgroup
constexpr int FC = 256;    // intermediate columns per chunk
constexpr int FBK = 64;    // k extent of a weight tile
constexpr int WLD = FBK + 8;   // LDS row stride of a weight tile (elements): 144-byte rows, conflict-free 16-byte reads
constexpr int ILD = FC + 8;    // row stride of the bf16 images / of the staged h tile
constexpr int W_STAGE_ELEMS = 256 * WLD;
constexpr int IMG_ELEMS = FBM * ILD;
constexpr int YLD = FD + 4;    // f32 row stride of the staged output tile
constexpr int FFN_MAX_F = 2048;
constexpr int FFN_SMEM = (2 * W_STAGE_ELEMS + 2 * IMG_ELEMS) * 2 + FFN_MAX_F * 4;  // weight stages + h image + u / a image + b1
static_assert(2 * W_STAGE_ELEMS * 2 >= FBM * YLD * 4, "the output tile is staged over the weight stages");

struct FfnArgs {
  int M, F;
  const bf16* x;
  const float *ln_g, *ln_b;
  float eps;
  const bf16* w1; const float* b1;   // [F, 256], [F]
  const bf16* w2; const float* b2;   // [256, F], [256]
  bf16 *h, *u, *a, *y;               // [M,256], [M,F] (u may be NULL), [M,F], [M,256]
  float *mean, *rstd;                // [M] (may be NULL)
  int act;
  float res_scale, drop_p;
  uint64_t seed_in, seed_out;
};

__device__ __forceinline__ void load4(const bf16* p, float (&o)[4]) {
  const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = (float)v[j];
}

__global__ __launch_bounds__(256, 1) void ffn_fwd_fused_kernel(const FfnArgs g) {
  using M_ = Mma<bf16>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* Ws0 = reinterpret_cast<bf16*>(smem);              // [2][256][WLD]
  bf16* Himg = Ws0 + 2 * W_STAGE_ELEMS;                   // [64][ILD]: h = LN(x) (B operand of the first product)
  bf16* Aimg = Himg + IMG_ELEMS;                          // [64][ILD]: pre-activation u (copied out), then the activation a
                                                          //            (copied out; A operand of the second product)
  float* B1s = reinterpret_cast<float*>(Aimg + IMG_ELEMS);  // [F]: the first bias (the mid epilogue reads it 16 bytes at a
                                                            //      time; from global memory each read was a serialised L2 trip)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, il = lane & 31, hh = lane >> 5;
  const int m0 = blockIdx.x * FBM;
  const int wm = (wave >> 1) * 32, wq = (wave & 1) * 128;  // rows of this wave; its 128 intermediate rows (U^T) / output columns

  for (int i = tid; i < g.F; i += 256) B1s[i] = g.b1[i];
  // ---- LayerNorm: wave w takes rows w, w + 4, ... of the block; one row = 64 lanes x 4 elements ---------------------
  for (int r = wave; r < FBM; r += 4) {
    const int row = m0 + r;
    float v[4] = {0.f, 0.f, 0.f, 0.f}, o[4] = {0.f, 0.f, 0.f, 0.f};
    if (row < g.M) {
      load4(g.x + (long)row * FD + lane * 4, v);
      const float mu = wave_sum(v[0] + v[1] + v[2] + v[3]) / FD;
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float dd = v[j] - mu; q += dd * dd; }
      const float rs = rsqrtf(wave_sum(q) / FD + g.eps);
      if (lane == 0) { if (g.mean) g.mean[row] = mu; if (g.rstd) g.rstd[row] = rs; }
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (v[j] - mu) * rs * g.ln_g[lane * 4 + j] + g.ln_b[lane * 4 + j];
    }
    bf16x4 ob;
#pragma unroll
    for (int j = 0; j < 4; ++j) ob[j] = (bf16)o[j];
    *reinterpret_cast<bf16x4*>(Himg + r * ILD + lane * 4) = ob;
    if (row < g.M) *reinterpret_cast<bf16x4*>(g.h + (long)row * FD + lane * 4) = ob;
  }
  __syncthreads();

  // ---- weight tile stream: tile q = chunk * 8 + s; s < 4: W1 rows [chunk*256, +256), k [64 s, +64);
  //                                                 s >= 4: W2 rows [0, 256), k = intermediate [chunk*256 + 64 (s-4), +64)
  const int nchunk = g.F / FC, nq = nchunk * 8;
  struct Stage { Vec16<bf16> v[8]; };
  const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(g.w1), rs2 = make_rsrc(g.w2);
  auto load_tile = [&](Stage& st, const int q) {
    const int c = q >> 3, s = q & 7;
    const bool on = q < nq;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = tid + 256 * i, row = p >> 3, kv = (p & 7) * 8;
      unsigned off;
      if (s < 4) off = (unsigned)((((c * FC + row) * FD) + s * FBK + kv) * 2);
      else off = (unsigned)(((row * g.F) + c * FC + (s - 4) * FBK + kv) * 2);
      st.v[i] = buf_load16<bf16>(s < 4 ? rs1 : rs2, on ? off : EMO_OOB);
    }
  };
  auto store_tile = [&](const Stage& st, const int buf) {
    bf16* W = Ws0 + buf * W_STAGE_ELEMS;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = tid + 256 * i;
      store16(W + (p >> 3) * WLD + (p & 7) * 8, st.v[i]);
    }
  };

  f32x16 accu[4], accy[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) { accu[j][r] = 0.f; accy[j][r] = 0.f; }

  // one tile: s < 4 -> U^T[f][m] += W1 tile . h^T   (A = W1 rows f, B = h rows m);  s >= 4 -> y[m][n] += a . W2^T
  auto compute = [&](const int q, const int buf) {
    const bf16* W = Ws0 + buf * W_STAGE_ELEMS;
    const int s = q & 7;
    if (s < 4) {
#pragma unroll
      for (int kk = 0; kk < FBK / 16; ++kk) {
        const bf16x8 hfr = M_::load_kc(Himg, ILD, wm, s * FBK + kk * 16, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) accu[j] = M_::mma(M_::load_kc(W, WLD, wq + j * 32, kk * 16, lane), hfr, accu[j]);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < FBK / 16; ++kk) {
        const bf16x8 af = M_::load_kc(Aimg, ILD, wm, (s - 4) * FBK + kk * 16, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) accy[j] = M_::mma(af, M_::load_kc(W, WLD, wq + j * 32, kk * 16, lane), accy[j]);
      }
    }
  };

  // bias + activation + dropout of the finished 256-column chunk, in registers; the image holds u first (copied out
  // row-major), then a (copied out, and read by the second product)
  auto copy_out = [&](bf16* dst, const int c) {  // 64 rows x 512-byte row segments of the image -> HBM, 16 bytes per lane
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = tid + 256 * i, r = p >> 5, cv = (p & 31) * 8;
      if (m0 + r < g.M)
        *reinterpret_cast<bf16x8*>(dst + (long)(m0 + r) * g.F + c * FC + cv) = *reinterpret_cast<const bf16x8*>(Aimg + r * ILD + cv);
    }
  };
  auto mid_epilogue = [&](const int c) {
    const int row = m0 + wm + il;                 // this lane's row (accumulator column)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int f = wq + j * 32 + 8 * g4 + 4 * hh;  // 4 consecutive intermediate columns: registers 4 g4 .. 4 g4 + 3
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(B1s + c * FC + f);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 1.f * accu[j][4 * g4 + e] + b4[e];
        if (g.u) {
          bf16x4 ub;
#pragma unroll
          for (int e = 0; e < 4; ++e) ub[e] = (bf16)v[e];
          *reinterpret_cast<bf16x4*>(Aimg + (wm + il) * ILD + f) = ub;
        }
        act_vec<4>(g.act, v);
        if (g.drop_p > 0.f) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] *= dropout_scale(g.seed_in, (uint64_t)row * (uint64_t)g.F + (uint64_t)(c * FC + f + e), g.drop_p);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) accu[j][4 * g4 + e] = v[e];  // parked until the image is free again
      }
    }
    if (g.u) {
      __syncthreads();
      copy_out(g.u, c);
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int f = wq + j * 32 + 8 * g4 + 4 * hh;
        bf16x4 ab;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ab[e] = (bf16)accu[j][4 * g4 + e]; accu[j][4 * g4 + e] = 0.f; }
        *reinterpret_cast<bf16x4*>(Aimg + (wm + il) * ILD + f) = ab;
      }
    __syncthreads();
    copy_out(g.a, c);
  };

  // 4-deep register ring over the flat tile sequence: tile t travels in st[t & 3] (loaded three steps before it is
  // multiplied) and is multiplied out of LDS[t & 1]; one barrier per tile.  The eight tiles of a chunk are unrolled, the
  // chunk loop is not: the loop body (with ONE copy of the mid epilogue) stays inside the instruction cache -- with the
  // steps instantiated three times over, each with its own epilogue, the kernel ran 2.6x slower than the unfused path.
  Stage st[4];
  load_tile(st[0], 0);
  load_tile(st[1], 1);
  load_tile(st[2], 2);
  store_tile(st[0], 0);
  __syncthreads();
#pragma unroll 1
  for (int c = 0; c < nchunk; ++c) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int q = c * 8 + s;
      load_tile(st[(s + 3) & 3], q + 3);
      compute(q, s & 1);
      if (s == 3) mid_epilogue(c);   // (its barrier orders the image writes before the reads of tiles s >= 4; the barrier
                                     //  below ends every read of the a image before the next chunk's epilogue)
      store_tile(st[(s + 1) & 3], (s + 1) & 1);
      __syncthreads();
    }
  }

  // ---- epilogue: y = x + res_scale * drop(acc + b2), staged f32 over the (now idle) weight stages -----------------
  float* sc = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[(wm + c_row(r, lane)) * YLD + wq + j * 32 + il] = accy[j][r];
  __syncthreads();
  {
    const int er = tid >> 5, ec = (tid & 31) * 8;  // 32 lanes cover one 256-column row, 8 rows per pass
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(g.b2 + ec);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(g.b2 + ec + 4);
#pragma unroll
    for (int pass = 0; pass < FBM / 8; ++pass) {
      const int lrow = pass * 8 + er, row = m0 + lrow;
      if (row < g.M) {
        float v[8];
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(&sc[lrow * YLD + ec]);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(&sc[lrow * YLD + ec + 4]);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = 1.f * v0[e] + b0[e]; v[4 + e] = 1.f * v1[e] + b1[e]; }
        if (g.drop_p > 0.f) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= dropout_scale(g.seed_out, (uint64_t)row * (uint64_t)FD + ec + e, g.drop_p);
        }
        float dres[8];
        load8<bf16>(g.x + (long)row * FD + ec, dres);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = dres[e] + g.res_scale * v[e];
        store8<bf16>(g.y + (long)row * FD + ec, v);
      }
    }
  }
}

}  // namespace

extern "C" int emoasr_ffn_fwd(int dtype, int M, int d, int F, const void* x, const float* ln_g, const float* ln_b, float eps,
                              const void* w1, const float* b1, const void* w2, const float* b2, int act, float res_scale,
                              float drop_p, uint64_t seed_in, uint64_t seed_out, void* h, float* mean, float* rstd, void* u,
                              void* a, void* y, void* stream) {
  EMO_CHECK(dtype == EMO_BF16 && d == FD && F % FC == 0 && F > 0, "ffn_fwd: fused form takes bf16, d = 256, F %% 256 == 0 (got d=%d F=%d)", d, F);
  EMO_CHECK(x && ln_g && ln_b && w1 && b1 && w2 && b2 && h && a && y, "ffn_fwd: missing arguments");
  EMO_CHECK((long)F * FD * 2 < (1L << 32), "ffn_fwd: weights too large for 32-bit buffer offsets");
  if (M == 0) return 0;
  FfnArgs g{};
  g.M = M; g.F = F; g.x = (const bf16*)x; g.ln_g = ln_g; g.ln_b = ln_b; g.eps = eps;
  g.w1 = (const bf16*)w1; g.b1 = b1; g.w2 = (const bf16*)w2; g.b2 = b2;
  g.h = (bf16*)h; g.u = (bf16*)u; g.a = (bf16*)a; g.y = (bf16*)y; g.mean = mean; g.rstd = rstd;
  g.act = act; g.res_scale = res_scale; g.drop_p = drop_p; g.seed_in = seed_in; g.seed_out = seed_out;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)ffn_fwd_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FFN_SMEM);
    if (e != hipSuccess) { emo_set_error("ffn_fwd: hipFuncSetAttribute(%d): %s", FFN_SMEM, hipGetErrorString(e)); return 1; }
    attr_set = true;
  }
  EMO_CHECK(F <= FFN_MAX_F, "ffn_fwd: F=%d > %d", F, FFN_MAX_F);
  ffn_fwd_fused_kernel<<<cdiv(M, FBM), 256, FFN_SMEM, (hipStream_t)stream>>>(g);
  EMO_LAUNCH_CHECK();
  return 0;
}
