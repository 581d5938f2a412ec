// The cached decode steps of config 4 (bf16, <= 16 live hypotheses) as ONE launch per network and step, spread over CG = 16
// co-resident workgroups that meet at grid barriers between the stages of a layer.
//
// Why: a step of either network is ~75 (decoder) / ~100 (LM) dependent launches of ~5 us each (csrc/decode_rt.hip; the chains
// are the step time: 0.68 ms at beam 10), while the work is one pass over 9 / 19 MB of weights for 10 rows.  A stage boundary
// INSIDE a launch -- every workgroup publishes its slice of the activations, one device-scope atomic add on a monotonic counter,
// a polling load, then 8 KB of activation rows read back -- measures 2.1 us for 16 workgroups on an otherwise idle device
// (tools/micro/grid_barrier.hip, profiles/r02_grid_barrier.txt: 1.8 / 2.1 / 2.8 / 4.4 us at 8 / 16 / 32 / 64 workgroups), and
// fusing LayerNorm, the cache append and the residual adds into the neighbouring stages leaves 5 (LM) / 8 (decoder; 7 with its projection and self-attention as one stage, see qkv_attn_stage) boundaries per
// layer instead of 8 / 12 launches.  The single-workgroup attempt (csrc/decode_wg.hip) had no boundaries at all but paid a global
// round trip per stage on one CU and could not stream the weights; here 16 workgroups x 8 waves each own up to four 16-column x
// 32-deep weight fragments of a stage, requested BEFORE the barrier that precedes the stage.
//
// MEASURED (MI355X, beam 10, d 256, F 1024 / 2048, rocprofv3, both chains running concurrently): LM stack 351 us per step (the
// launch chain: ~500), decoder stack 339 us (~375; 312 with its projection + self-attention merged); search step 0.68 -> 0.47-0.58
// ms over T' 190-600.  In-kernel stamps (-DEMO_COOP_STAMP; s_memtime
// ticks at the 2.39 GHz shader clock, tools/micro/clock_probe.hip) of the LM stack, per layer: barriers 5 x 2.0 us (gather 0.25,
// release fence 0.65, add 0.1, wait 0.8, acquire fence 0.2), QKV stage 3.8 us (of which LayerNorm 1.6), attention 4.5 (q load 0.8,
// one 64-key pass 3.2), out-projection 1.9, FFN 3.8 + 2.75: ~27 us per layer.  What it took to get there: 512 instead of 1024
// threads (at 128 registers the kernels spilled, and a spill reloaded after the barrier's cache invalidate is a memory round
// trip), per-stage index arithmetic kept out of the layer loop's preheader (EMO_FRESH), the layers' parameter pointers in LDS
// instead of the kernel-argument segment, release / acquire fences instead of two full fences per barrier (480 -> 351 us
// together); DPP instead of shuffle reductions changed little.  A stage is a handful of dependent memory round trips (~0.8 us each
// after the invalidate) and dependent-instruction chains (5.75 cycles per dependent VALU op) on a nearly idle chip: the remaining
// lever is fewer stages, not faster ones.
//
//   linear stage  : all activation rows [16][K] -> LDS (every workgroup; LayerNorm recomputed by each, one wave per row);
//                   workgroup g owns the 16-column strips g, g + 16, ...; its 16 waves split (strip, k step) units, one 16x16x32
//                   MFMA per unit, partial tiles summed through LDS; epilogue (bias, activation, residual) by one thread per output
//   self-attention: the new key / value go straight from the projection's epilogue into the caches; one (hypothesis, head) pair
//                   per wave, a lane per key for the scores, a lane per pair of output dimensions for P.V with the half-waves
//                   taking the even / odd keys (all loads of a 64-key pass in flight together)
//   source attn.  : split over 64-key chunks of the encoder memory: (pair, chunk) items over all 256 waves, partial (max, sum,
//                   out[dk]) to global memory, merged by the next stage while it loads its rows
//
// Memory model: the activations travel through device memory between workgroups on different XCDs (whose L2s are not coherent
// for ordinary memory): the barrier's arrive is preceded by an agent-scope release fence (L2 write-back) and its wait is followed
// by an agent-scope acquire fence (L1 / L2 invalidate), executed by thread 0 between two __syncthreads().  A wait that does not complete
// within ~2^22 polls (a lost workgroup, i.e. a bug or a device that cannot hold 16 workgroups at once) sets an error flag and falls
// through instead of hanging the device.
//
// Reference: decoders/transformer.py:148-159 + transformer.py:156-198 (pre-LN decoder layer, ReLU);
// lm/modeling/transformer.py:62-77 over modeling_bert.py:159-303,360-436 (post-LN block, GELU).
#include <math.h>
#include "common.h"
#include "../../include/emoasr_hip.h"

namespace {

constexpr int CG = 16;          // workgroups per network (a power of two: the counter's wrap-around keeps its phase)
constexpr int CT = 512;         // threads per workgroup: 8 waves = 2 per SIMD, 256 registers each (at 1024 threads / 128 registers
                                // the kernels spilled, and every spill reload after a barrier's cache invalidate is a memory round trip)
constexpr int CW = CT / 64;     // waves
constexpr int C_MAXD = 512;     // widest model dimension (LayerNorm rows: 8 values per lane)
constexpr int C_MAXF = 1024;    // widest activation row
constexpr int C_LD = C_MAXF + 8;
constexpr int C_MAXPAIR = 64;   // hypotheses x heads
constexpr int C_MAXCHUNK = 32;  // 64-key chunks of the encoder memory (T' <= 2048)

struct CoopBufs {               // device scratch of one chain (allocated once, see coop_bufs())
  unsigned* counter;            // monotonic arrivals
  int* err;
  bf16 *X, *Y, *O, *QKV, *ACT, *Q2;   // [16][d] x3, [16][3d], [16][F], [16][d]
  float* PART;                  // [pairs][chunks][dk + 2]
};

struct LmCoopArgs {
  int nl, nb, Lmax, d, H, F;
  const int* ids; const int* pos;
  const bf16* word_emb; const float* pe; emoasr_lnp_t ln_emb;
  bf16* kcache; bf16* vcache;
  emoasr_lin_t transform;
  bf16* out_hidden;             // [nb, d]: GELU(transform(x)); the head's LayerNorm + tied projection follow in emoasr_rowlin
  CoopBufs B;
  emoasr_bert_layer_t layers[12];
};
struct DecCoopArgs {
  int nl, nb, Lmax, T, d, H, F;
  const int* ids; const int* pos;
  const bf16* embed; const float* pe; float emb_scale;
  bf16* kcache; bf16* vcache;
  const void* kv[8];            // cross-attention K | V of the encoder memory per layer: [nb][T][2d]
  const int* kmem;
  bf16* out_x;                  // [nb, d]: the stack's output before the final LayerNorm
  CoopBufs B;
  emoasr_decoder_layer_t layers[8];
};

// Wave-wide reductions through DPP moves (quad swaps, half-row and row mirrors, then the row broadcasts of GFX9): ~7 short
// instructions each; the shuffle form (__shfl_xor = ds_bpermute, ~100 cycles of LDS-crossbar latency per step, 6 dependent steps)
// made a LayerNorm of two rows per wave cost 2.5 us and the soft-max of a 64-key attention pass most of its 4 us.
#define EMO_DPP_F(old_, v_, ctrl_, rmask_) \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(old_)), __builtin_bit_cast(int, (v_)), (ctrl_), (rmask_), 0xf, false))
__device__ __forceinline__ float wave_sum(float v) {
  v += EMO_DPP_F(0.f, v, 0xB1, 0xf);    // quad_perm [1,0,3,2]
  v += EMO_DPP_F(0.f, v, 0x4E, 0xf);    // quad_perm [2,3,0,1]
  v += EMO_DPP_F(0.f, v, 0x141, 0xf);   // row_half_mirror
  v += EMO_DPP_F(0.f, v, 0x140, 0xf);   // row_mirror: every lane holds its row's (16 lanes) sum
  v += EMO_DPP_F(0.f, v, 0x142, 0xa);   // row_bcast15 into rows 1, 3
  v += EMO_DPP_F(0.f, v, 0x143, 0xc);   // row_bcast31 into rows 2, 3: lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, EMO_DPP_F(v, v, 0xB1, 0xf));
  v = fmaxf(v, EMO_DPP_F(v, v, 0x4E, 0xf));
  v = fmaxf(v, EMO_DPP_F(v, v, 0x141, 0xf));
  v = fmaxf(v, EMO_DPP_F(v, v, 0x140, 0xf));
  v = fmaxf(v, EMO_DPP_F(v, v, 0x142, 0xa));
  v = fmaxf(v, EMO_DPP_F(v, v, 0x143, 0xc));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// sum over the lane's HALF of the wave (lanes 0-31 / 32-63), both halves at once
__device__ __forceinline__ float half_sum(float v, int lane) {
  v += EMO_DPP_F(0.f, v, 0xB1, 0xf);
  v += EMO_DPP_F(0.f, v, 0x4E, 0xf);
  v += EMO_DPP_F(0.f, v, 0x141, 0xf);
  v += EMO_DPP_F(0.f, v, 0x140, 0xf);
  v += EMO_DPP_F(0.f, v, 0x142, 0xa);   // rows 1, 3 now hold the sums of lanes 0-31 / 32-63
  const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
  const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
  return lane < 32 ? lo : hi;
}
__device__ __forceinline__ float gelu_(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

// Thread-index arithmetic of a stage is recomputed where the stage runs: hoisted out of the layer loop (it is loop-invariant) the
// dozens of per-stage offsets outlive the registers, and a spilled value reloaded right after a barrier's cache invalidate costs a
// memory round trip.
#define EMO_FRESH(x) asm volatile("" : "+v"(x))

// ---- grid barrier ------------------------------------------------------------------------------------------------------------
struct Bar { unsigned* counter; int* err; unsigned base, n; int chain, ns; };
#ifdef EMO_COOP_STAMP
__device__ unsigned long long g_coop_stamps[2][8192];
__device__ int g_coop_nstamp[2];
__device__ int g_stamp_n;   // (per-thread copy lives in a register: see CSTAMP)
#define CSTAMP(b, id)                                                                              \
  do {                                                                                             \
    if (blockIdx.x == 0 && threadIdx.x == 0 && (b).ns < 8190) {                                    \
      g_coop_stamps[(b).chain][(b).ns++] = (__builtin_amdgcn_s_memtime() << 8) | (unsigned)(id);   \
      g_coop_nstamp[(b).chain] = (b).ns;                                                           \
    }                                                                                              \
  } while (0)
#else
#define CSTAMP(b, id) do {} while (0)
#endif

// Every launch performs the same number of barriers with CG arrivals each and launches of one chain are stream-ordered, so the
// counter is a multiple of CG when a launch starts; a workgroup that starts late sees at most CG - 1 arrivals of the FIRST barrier
// on top of that (the barrier cannot complete without it), so rounding down recovers the launch's base on every workgroup.
__device__ void bar_init(Bar& b, unsigned* counter, int* err, unsigned* s_base) {
  if (threadIdx.x == 0) {
    const unsigned v = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_base = v - v % (unsigned)CG;
  }
  __syncthreads();
  b.counter = counter; b.err = err; b.base = *s_base; b.n = 0; b.chain = 0; b.ns = 0;
}
__device__ void grid_sync(Bar& b, int id = 1) {
  ++b.n;
  CSTAMP(b, id);   // stage work done
  __syncthreads();
  CSTAMP(b, 2);   // workgroup gathered
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // L2 write-back only (a full __threadfence() also invalidates, twice per barrier)
    CSTAMP(b, 3); // released
    __hip_atomic_fetch_add(b.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    CSTAMP(b, 4); // arrived
    const unsigned target = b.base + (unsigned)CG * b.n;
    unsigned spins = 0;
    while ((int)(__hip_atomic_load(b.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22)) { *b.err = 1; break; }
    }
    CSTAMP(b, 5); // everybody arrived
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // L1 / L2 invalidate
    CSTAMP(b, 6); // acquired
  }
  __syncthreads();
}

// ---- linear stages -----------------------------------------------------------------------------------------------------------
// Work of wave `wave` of workgroup `g` in y[16][N] = x[16][K] . W[N][K]^T: the workgroup's ns strips x (K / 32) k steps are dealt
// to the waves in runs of upw (1, 2 or 4) consecutive k steps of one strip.
constexpr int C_UPW = 4;
struct Plan { int ns, tps, sl, k0, cnt; };
__device__ __forceinline__ Plan make_plan(int N, int K, int g, int wave) {
  Plan p;
  const int nstrip = (N + 15) / 16;
  p.ns = nstrip > g ? (nstrip - g + CG - 1) / CG : 0;
  const int ksteps = K / 32, total = p.ns * ksteps;
  const int upw = total > 2 * CW ? 4 : (total > CW ? 2 : 1);
  p.tps = ksteps / upw;                 // partial tiles per strip
  const int u0 = wave * upw;
  p.cnt = u0 < total ? min(upw, total - u0) : 0;
  p.sl = p.cnt ? u0 / ksteps : 0;
  p.k0 = (u0 % ksteps) * 32;
  return p;
}
constexpr int C_OPT = 1024 / CT;   // outputs per thread of a linear stage (at most 4 strips x 256 per workgroup)
struct WFrag { bf16x8 w[C_UPW]; float bias[C_OPT]; };
template <int NV> struct LnFragT { float g[NV], b[NV]; };   // NV = d / 64 values per lane (4 where d <= 256 is known, else 8)

// a pointer read from LDS is uniform but sits in a vector register: buffer descriptors want it in scalar ones
template <typename P>
__device__ __forceinline__ P* uniform_ptr(P* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (P*)(((unsigned long long)hi << 32) | lo);
}

// request this thread's share of a linear stage's parameters (no dependence on the activations: issued ahead of the barrier)
__device__ __forceinline__ void wprefetch(WFrag& f, const emoasr_lin_t& L, int N, int K, int g, int tid) {
  EMO_FRESH(tid);
  const int wave = tid >> 6, lane = tid & 63;
  const Plan p = make_plan(N, K, g, wave);
  const void* Lw = uniform_ptr(L.w);
  const float* Lb = uniform_ptr(L.b);
  const __amdgpu_buffer_rsrc_t rsw = make_rsrc(Lw), rsb = make_rsrc(Lb);
  const int col = (g + CG * p.sl) * 16 + (lane & 15);
  const bool cok = p.cnt > 0 && col < N;
  const unsigned woff = (unsigned)(((long)(cok ? col : 0) * K + p.k0 + 8 * (lane >> 4)) * 2);
#pragma unroll
  for (int i = 0; i < C_UPW; ++i) f.w[i] = buf_load16<bf16>(rsw, (cok && i < p.cnt) ? woff + (unsigned)(64 * i) : EMO_OOB).v;
#pragma unroll
  for (int o = 0; o < C_OPT; ++o) {   // the outputs this thread finishes: strip (tid + o CT) >> 8, column tid & 15
    const int sl = (tid + o * CT) >> 8, ocol = (g + CG * sl) * 16 + (tid & 15);
    f.bias[o] = buf_load_f32<float>(rsb, (sl < p.ns && ocol < N && Lb) ? (unsigned)(ocol * 4) : EMO_OOB);
  }
}
// likewise the LayerNorm parameters of the rows a stage normalises: lane owns columns lane, lane + 64, ...
template <int NV>
__device__ __forceinline__ void lnprefetch(LnFragT<NV>& f, const emoasr_lnp_t& ln, int d, int tid) {
  EMO_FRESH(tid);
  const int lane = tid & 63;
  const __amdgpu_buffer_rsrc_t rsg = make_rsrc(uniform_ptr(ln.g)), rsb = make_rsrc(uniform_ptr(ln.b));
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int k = lane + 64 * i;
    f.g[i] = buf_load_f32<float>(rsg, k < d ? (unsigned)(k * 4) : EMO_OOB);
    f.b[i] = buf_load_f32<float>(rsb, k < d ? (unsigned)(k * 4) : EMO_OOB);
  }
}
// the residual elements matching this thread's outputs (res[16][ld], the previous stages' rows in global memory)
struct ResFrag { float v[C_OPT]; };
__device__ __forceinline__ void resfetch(ResFrag& r, const bf16* res, int ld, int N, int g, int tid) {
  EMO_FRESH(tid);
#pragma unroll
  for (int o = 0; o < C_OPT; ++o) {
    const int sl = (tid + o * CT) >> 8, n = (g + CG * sl) * 16 + (tid & 15), m = (tid & 255) >> 4;
    r.v[o] = n < N ? (float)res[(long)m * ld + n] : 0.f;
  }
}

// epi(m, n, v, o) for this workgroup's outputs, v = bias[n] + sum_k xs[m][k] W[n][k]; o = the thread's output slot (ResFrag index)
template <typename Epi>
__device__ __forceinline__ void coop_linear(const WFrag& f, const bf16* xs, int N, int K, int g, int tid, float* red, Epi epi) {
  EMO_FRESH(tid);
  const int wave = tid >> 6, lane = tid & 63;
  const Plan p = make_plan(N, K, g, wave);
  if (p.cnt > 0) {
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16* xrow = xs + (lane & 15) * C_LD + p.k0 + 8 * (lane >> 4);
#pragma unroll
    for (int i = 0; i < C_UPW; ++i)
      if (i < p.cnt) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(xrow + 32 * i), f.w[i], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave * 256 + (4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[r];
  }
  __syncthreads();
#pragma unroll
  for (int o = 0; o < C_OPT; ++o) {
    const int sl = (tid + o * CT) >> 8, idx = tid & 255;
    if (sl < p.ns) {
      const int n = (g + CG * sl) * 16 + (idx & 15);
      if (n < N) {
        float v = f.bias[o];
        for (int j = 0; j < p.tps; ++j) v += red[(sl * p.tps + j) * 256 + idx];
        epi(idx >> 4, n, v, o);
      }
    }
  }
  __syncthreads();   // `red` and the rows may be rewritten
}

// ---- activation rows ---------------------------------------------------------------------------------------------------------
// all 16 rows of src[16][ncols] (bf16, dense) -> rows[16][C_LD]
__device__ __forceinline__ void load_rows(bf16* rows, const bf16* src, int ncols, int tid) {
  EMO_FRESH(tid);
  const int per_row = ncols / 8;
  for (int i = tid; i < 16 * per_row; i += CT) {
    const int m = i / per_row, c = (i - m * per_row) * 8;
    *reinterpret_cast<bf16x8*>(rows + m * C_LD + c) = *reinterpret_cast<const bf16x8*>(src + (long)m * ncols + c);
  }
}
// rows[r][0..d) <- LayerNorm(rows[r][0..d)), one row per HALF-wave (CW = 8: all 16 rows in one pass; two rows per wave one after
// the other cost 1.6 us per LayerNorm), a lane owns columns hl, hl + 32, ... (eps 1e-12, as both networks use); the caller
// synchronises.  P: gamma / beta of columns lane + 64 i as lnprefetch left them -- a half's lane hl needs hl + 32 j, which the lane
// itself (j even) or its partner in the other half (j odd) holds: fetched with one cross-half swap per value.
template <int NV>
__device__ __forceinline__ void ln_rows(bf16* rows, int d, const LnFragT<NV>& P, int tid) {
  EMO_FRESH(tid);
  static_assert(CW == 8, "one row per half-wave covers 16 rows with 8 waves");
  const int wave = tid >> 6, lane = tid & 63, hl = lane & 31, half = lane >> 5;
  const int r = 2 * wave + half;
  float v[2 * NV], gg[2 * NV], bb[2 * NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    // columns 64 i + hl (held by lane hl) and 64 i + 32 + hl (held by lane 32 + hl): both halves need both
    const float g_own = P.g[i], b_own = P.b[i];
    const float g_oth = __shfl_xor(g_own, 32), b_oth = __shfl_xor(b_own, 32);
    gg[2 * i] = half ? g_oth : g_own;      bb[2 * i] = half ? b_oth : b_own;
    gg[2 * i + 1] = half ? g_own : g_oth;  bb[2 * i + 1] = half ? b_own : b_oth;
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 2 * NV; ++j) {
    const int k = hl + 32 * j;
    v[j] = k < d ? (float)rows[r * C_LD + k] : 0.f;
    s += v[j];
  }
  const float mean = half_sum(s, lane) / d;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < 2 * NV; ++j) {
    const float dl = hl + 32 * j < d ? v[j] - mean : 0.f;
    q += dl * dl;
  }
  const float rstd = rsqrtf(half_sum(q, lane) / d + 1e-12f);
#pragma unroll
  for (int j = 0; j < 2 * NV; ++j) {
    const int k = hl + 32 * j;
    if (k < d) rows[r * C_LD + k] = (bf16)((v[j] - mean) * rstd * gg[j] + bb[j]);
  }
}
// rows -> dst[16][d] (workgroup 0 publishes the normalised rows: the post-LN blocks add them back two stages later)
__device__ __forceinline__ void store_rows(bf16* dst, const bf16* rows, int d, int tid) {
  EMO_FRESH(tid);
  const int per_row = d / 8;
  for (int i = tid; i < 16 * per_row; i += CT) {
    const int m = i / per_row, c = (i - m * per_row) * 8;
    *reinterpret_cast<bf16x8*>(dst + (long)m * d + c) = *reinterpret_cast<const bf16x8*>(rows + m * C_LD + c);
  }
}

// ---- attention ----------------------------------------------------------------------------------------------------------------
// One (hypothesis, head) pair x one run of <= 64 keys on one wave (dk <= 64): scores with a lane per key; P.V with a lane per
// PAIR of output dimensions and the two half-waves taking the even / odd keys (32 four-byte loads per lane instead of 64 two-byte
// ones: the kernel runs at 128 registers per thread), folded with one cross-half add at the end.  Every global load of the pass
// is issued before the first is used.  Returns the un-normalised result: m (max score), l (sum of exp), and on lanes < dk / 2 the
// output dimensions 2 lane (o0) and 2 lane + 1 (o1).  q: dk floats in wave-private LDS, prob: 64 floats.
__device__ __forceinline__ void attend64(const float* q, int dk, const bf16* kbase, long kstride, const bf16* vbase, long vstride,
                                         int nkeys, float scale, float* prob, int lane, float& m_out, float& l_out, float& o0,
                                         float& o1) {
  const __amdgpu_buffer_rsrc_t rsk = make_rsrc(kbase), rsv = make_rsrc(vbase);
  const bool tok = lane < nkeys;
  const int nch = dk / 8, dp = lane & 31, half = lane >> 5;
  bf16x8 kv[8];
#pragma unroll
  for (int c = 0; c < 8; ++c)
    kv[c] = buf_load16<bf16>(rsk, (tok && c < nch) ? (unsigned)(((long)lane * kstride + 8 * c) * 2) : EMO_OOB).v;
  unsigned vv[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int j = 2 * i + half;
    vv[i] = __builtin_amdgcn_raw_buffer_load_b32(rsv, (j < nkeys && 2 * dp < dk) ? (unsigned)(((long)j * vstride + 2 * dp) * 2) : EMO_OOB, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c < nch) {
      const f32x4 qa = *reinterpret_cast<const f32x4*>(q + 8 * c), qb = *reinterpret_cast<const f32x4*>(q + 8 * c + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) s += qa[e] * (float)kv[c][e];
#pragma unroll
      for (int e = 0; e < 4; ++e) s += qb[e] * (float)kv[c][4 + e];
    }
  }
  s = tok ? s * scale : -INFINITY;
  const float m = wave_max(s);
  const float p = tok ? __expf(s - m) : 0.f;
  prob[(lane & 1) * 32 + (lane >> 1)] = p;   // [parity][key >> 1] (wave-private LDS: visible to this wave's reads below in program order)
  l_out = wave_sum(p);
  m_out = m;
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int i4 = 0; i4 < 8; ++i4) {
    const f32x4 pj = *reinterpret_cast<const f32x4*>(prob + half * 32 + 4 * i4);   // keys 2 (4 i4 + e) + half
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a0 += pj[e] * __uint_as_float(vv[4 * i4 + e] << 16);           // bf16 -> f32: the low half is dimension 2 dp
      a1 += pj[e] * __uint_as_float(vv[4 * i4 + e] & 0xFFFF0000u);
    }
  }
  o0 = a0 + __shfl_xor(a0, 32);
  o1 = a1 + __shfl_xor(a1, 32);
}

// self-attention over the cached prefix (keys 0 .. pos; the projection stage appended row pos): pair -> workgroup pair % CG,
// wave pair / CG; out rows [16][d] in global memory
__device__ void coop_self_attention(const bf16* qkv, int nb, int d, int H, int Lmax, const bf16* kc, const bf16* vc, int pos,
                                    bf16* out, int g, int tid, float* wscr, Bar& bar) {
  EMO_FRESH(tid);
  CSTAMP(bar, 26);
  const int wave = tid >> 6, lane = tid & 63, dk = d / H;
  float* q = wscr + wave * 192;
  float* prob = q + 128;
  for (int pair = g + CG * wave; pair < nb * H; pair += CG * CW) {
    const int b = pair / H, hh = pair - b * H;
    for (int c = lane; c < dk; c += 64) q[c] = (float)qkv[(long)b * 3 * d + hh * dk + c];
    CSTAMP(bar, 27);
    float m_run = -INFINITY, l_run = 0.f, a0 = 0.f, a1 = 0.f;
    const bf16* kb = kc + (long)b * Lmax * d + hh * dk;
    const bf16* vb = vc + (long)b * Lmax * d + hh * dk;
    for (int t0 = 0; t0 <= pos; t0 += 64) {
      float m, l, o0, o1;
      attend64(q, dk, kb + (long)t0 * d, d, vb + (long)t0 * d, d, min(64, pos + 1 - t0), 1.f / sqrtf((float)dk), prob, lane, m, l, o0, o1);
      const float mn = fmaxf(m_run, m), ca = __expf(m_run - mn), cb = __expf(m - mn);
      l_run = l_run * ca + l * cb; a0 = a0 * ca + o0 * cb; a1 = a1 * ca + o1 * cb;
      m_run = mn;
      CSTAMP(bar, 28);
    }
    const float inv = 1.f / l_run;
    if (2 * lane < dk) {
      out[(long)b * d + hh * dk + 2 * lane] = (bf16)(a0 * inv);
      out[(long)b * d + hh * dk + 2 * lane + 1] = (bf16)(a1 * inv);
    }
  }
}

// ---- projection + self-attention in one stage (no grid barrier between them) ------------------------------------------------------
// Workgroup g takes head g % H (CG / H workgroups per head): each of them computes that head's q | k | v columns for all 16 rows
// -- 3 dk / 16 strips, a wave owns one or two whole strips over every k step, so no cross-wave sum -- into LDS, and then attends
// for the hypotheses b = g / H, g / H + CG / H, ...  The head's new key / value rows come from LDS, the older ones from the caches;
// the first workgroup of a head appends the new rows to the caches for the later steps.  The projection is computed CG / H times
// over, which costs less than the barrier and the q round trip it replaces (2.8 us per layer).
constexpr int C_QKS = 8;   // k steps of the projection (d <= 256)
constexpr int C_QPRE = 4;  // ... of which this many (first strip) are requested ahead of the barrier
struct QFrag { bf16x8 w[2][C_QKS]; float bias[2]; };
__host__ __device__ inline bool qkv_merge_ok(int d, int H) {
  const int dk = d / H;
  return CG % H == 0 && dk % 16 == 0 && 3 * dk / 16 <= 2 * CW && d / 32 <= C_QKS && dk <= 64;
}
// part 0: the first C_QPRE k steps of the wave's first strip (requested ahead of the barrier); part 1: the rest, requested at the
// start of the stage -- its round trip hides behind the LayerNorm -- so that only 16 registers of fragments live across the
// barrier (all 64 did not fit next to the LM kernel's other state: 28 were spilled and reloaded, a memory round trip each, right
// before their MFMAs)
__device__ __forceinline__ void qprefetch(QFrag& f, const emoasr_lin_t& L, int d, int H, int g, int tid, int part) {
  EMO_FRESH(tid);
  const int wave = tid >> 6, lane = tid & 63, dk = d / H, h = g % H, spp = dk / 16, nstr = 3 * spp, ksteps = d / 32;
  const void* Lw = uniform_ptr(L.w);
  const float* Lb = uniform_ptr(L.b);
  const __amdgpu_buffer_rsrc_t rsw = make_rsrc(Lw), rsb = make_rsrc(Lb);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int st = wave + CW * i;
    const bool ok = st < nstr;
    const int col = (st / spp) * d + h * dk + (st % spp) * 16 + (lane & 15);
    const unsigned woff = (unsigned)(((long)(ok ? col : 0) * d + 8 * (lane >> 4)) * 2);
#pragma unroll
    for (int ks = 0; ks < C_QKS; ++ks) {
      const bool early = i == 0 && ks < C_QPRE;   // part 0: the first k steps of the first strip
      if (early == (part == 0)) f.w[i][ks] = buf_load16<bf16>(rsw, (ok && ks < ksteps) ? woff + (unsigned)(64 * ks) : EMO_OOB).v;
    }
    if (part == 1) f.bias[i] = buf_load_f32<float>(rsb, (ok && Lb) ? (unsigned)(col * 4) : EMO_OOB);
  }
}
// rows: the normalised input rows [16][C_LD]; hq: [16][3 dk + 8] bf16 scratch (the `red` buffer); out: O rows [16][d] in global memory
__device__ void qkv_attn_stage(const QFrag& f, const bf16* rows, bf16* hq, int nb, int d, int H, int Lmax, bf16* kc, bf16* vc, int pos,
                               bf16* out, int g, int tid, float* wscr) {
  EMO_FRESH(tid);
  const int wave = tid >> 6, lane = tid & 63, dk = d / H, h = g % H, slot = g / H, wph = CG / H;
  const int spp = dk / 16, nstr = 3 * spp, ksteps = d / 32, ldh = 3 * dk + 8;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int st = wave + CW * i;
    if (st < nstr) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      const bf16* xrow = rows + (lane & 15) * C_LD + 8 * (lane >> 4);
#pragma unroll
      for (int ks = 0; ks < C_QKS; ++ks)
        if (ks < ksteps) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(xrow + 32 * ks), f.w[i][ks], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) hq[(4 * (lane >> 4) + r) * ldh + st * 16 + (lane & 15)] = (bf16)(acc[r] + f.bias[i]);   // [q | k | v] of the head
    }
  }
  __syncthreads();
  if (slot == 0)   // the new key / value rows of this head -> caches (read by the later steps)
    for (int i = tid; i < nb * dk; i += CT) {
      const int m = i / dk, c = i - m * dk;
      kc[((long)m * Lmax + pos) * d + h * dk + c] = hq[m * ldh + dk + c];
      vc[((long)m * Lmax + pos) * d + h * dk + c] = hq[m * ldh + 2 * dk + c];
    }
  float* q = wscr + wave * 192;
  float* prob = q + 128;
  const float scale = 1.f / sqrtf((float)dk);
  for (int b = slot + wph * wave; b < nb; b += wph * CW) {
    for (int c = lane; c < dk; c += 64) q[c] = (float)hq[b * ldh + c];
    float m_run = -INFINITY, l_run = 0.f, a0 = 0.f, a1 = 0.f;
    const bf16* kb = kc + (long)b * Lmax * d + h * dk;
    const bf16* vb = vc + (long)b * Lmax * d + h * dk;
    for (int t0 = 0; t0 < pos; t0 += 64) {   // the cached keys 0 .. pos - 1
      float m, l, o0, o1;
      attend64(q, dk, kb + (long)t0 * d, d, vb + (long)t0 * d, d, min(64, pos - t0), scale, prob, lane, m, l, o0, o1);
      const float mn = fmaxf(m_run, m), ca = __expf(m_run - mn), cb = __expf(m - mn);
      l_run = l_run * ca + l * cb; a0 = a0 * ca + o0 * cb; a1 = a1 * ca + o1 * cb;
      m_run = mn;
    }
    {  // the new key (position pos) out of LDS
      const float sn = wave_sum(lane < dk ? q[lane] * (float)hq[b * ldh + dk + lane] : 0.f) * scale;
      const float mn = fmaxf(m_run, sn), ca = __expf(m_run - mn), cb = __expf(sn - mn);
      const float v0 = 2 * lane < dk ? (float)hq[b * ldh + 2 * dk + 2 * lane] : 0.f;
      const float v1 = 2 * lane < dk ? (float)hq[b * ldh + 2 * dk + 2 * lane + 1] : 0.f;
      l_run = l_run * ca + cb; a0 = a0 * ca + v0 * cb; a1 = a1 * ca + v1 * cb;
    }
    const float inv = 1.f / l_run;
    if (2 * lane < dk) {
      out[(long)b * d + h * dk + 2 * lane] = (bf16)(a0 * inv);
      out[(long)b * d + h * dk + 2 * lane + 1] = (bf16)(a1 * inv);
    }
  }
  __syncthreads();   // hq (= red) and the rows may be rewritten
}

// ---- the Transformer LM stack ----------------------------------------------------------------------------------------------------
template <bool MERGED>
__global__ __launch_bounds__(CT) void lm_step_coop_kernel(const LmCoopArgs a) {
  __shared__ __attribute__((aligned(16))) bf16 rows[16 * C_LD];
  __shared__ float red[4096];
  __shared__ __attribute__((aligned(16))) float wscr[CW * 192];
  __shared__ unsigned s_base;
  const int tid = threadIdx.x, g = blockIdx.x;
  const int d = a.d, nb = a.nb, F = a.F, pos = *a.pos;
  const CoopBufs& B = a.B;
  Bar bar;
  bar_init(bar, B.counter, B.err, &s_base);
  bar.chain = 1;
  WFrag wf;
  LnFragT<MERGED ? 4 : C_MAXD / 64> lf;   // (the merged stage implies d <= 256)
  ResFrag rf;
  // the layers' parameter pointers -> LDS once: read from the kernel-argument segment where they are needed, every stage began
  // with a scalar load that the previous barrier's invalidate had turned into a memory round trip
  __shared__ emoasr_bert_layer_t s_layers[12];
  for (int i = tid; i < (int)(sizeof(emoasr_bert_layer_t) / 4) * a.nl; i += CT)
    reinterpret_cast<unsigned*>(s_layers)[i] = reinterpret_cast<const unsigned*>(a.layers)[i];
  __syncthreads();
  constexpr bool merged = MERGED;   // projection + self-attention in one stage (the host checks qkv_merge_ok)
  QFrag qf;
  // embeddings (word + position / token-type) -> rows; rows >= nb are zero
  for (int i = tid; i < 16 * d; i += CT) {
    const int m = i / d, c = i - m * d;
    rows[m * C_LD + c] = m < nb ? (bf16)((float)a.word_emb[(long)a.ids[m] * d + c] + a.pe[(long)pos * d + c]) : (bf16)0.f;
  }
  __syncthreads();
  if constexpr (merged) qprefetch(qf, s_layers[0].qkv, d, a.H, g, tid, 0);
  else wprefetch(wf, s_layers[0].qkv, 3 * d, d, g, tid);
  lnprefetch(lf, a.ln_emb, d, tid);
  const long layer_elems = (long)nb * a.Lmax * d;
  for (int li = 0; li < a.nl; ++li) {
    const emoasr_bert_layer_t& Ly = s_layers[li];
    bf16* kc = a.kcache + li * layer_elems;
    bf16* vc = a.vcache + li * layer_elems;
    // S1: x = LN(rows) (the previous block's output LayerNorm / the embedding LayerNorm); q | k | v = x . Wqkv^T + b
    if constexpr (merged) qprefetch(qf, Ly.qkv, d, a.H, g, tid, 1);
    ln_rows(rows, d, lf, tid);
    __syncthreads();
    CSTAMP(bar, 16);
    if (g == 0) store_rows(B.X, rows, d, tid);
    if constexpr (merged) {
      // S1 + S2: the head's q | k | v and its attention over keys 0 .. pos, no barrier in between
      qkv_attn_stage(qf, rows, reinterpret_cast<bf16*>(red), nb, d, a.H, a.Lmax, kc, vc, pos, B.O, g, tid, wscr);
      wprefetch(wf, Ly.attn_out, d, d, g, tid);
    } else {
      coop_linear(wf, rows, 3 * d, d, g, tid, red, [&](int m, int n, float v, int) {
        if (n < d) B.QKV[(long)m * 3 * d + n] = (bf16)v;
        else if (m < nb) {
          if (n < 2 * d) kc[((long)m * a.Lmax + pos) * d + n - d] = (bf16)v;
          else vc[((long)m * a.Lmax + pos) * d + n - 2 * d] = (bf16)v;
        }
      });
      CSTAMP(bar, 17);
      wprefetch(wf, Ly.attn_out, d, d, g, tid);
      grid_sync(bar, 21);
      // S2: attention over keys 0 .. pos
      coop_self_attention(B.QKV, nb, d, a.H, a.Lmax, kc, vc, pos, B.O, g, tid, wscr, bar);
    }
    lnprefetch(lf, Ly.ln_attn, d, tid);
    grid_sync(bar, 22);
    // S3: y = o . Wout^T + b + x
    CSTAMP(bar, 10);
    resfetch(rf, B.X, d, d, g, tid);
    load_rows(rows, B.O, d, tid);
    CSTAMP(bar, 11);
    __syncthreads();
    CSTAMP(bar, 12);
    coop_linear(wf, rows, d, d, g, tid, red, [&](int m, int n, float v, int o) { B.Y[(long)m * d + n] = (bf16)(v + rf.v[o]); });
    CSTAMP(bar, 13);
    wprefetch(wf, Ly.inter, F, d, g, tid);
    CSTAMP(bar, 14);
    grid_sync(bar, 23);
    // S4: x' = LN(y); act = GELU(x' . W1^T + b1)
    load_rows(rows, B.Y, d, tid);
    __syncthreads();
    ln_rows(rows, d, lf, tid);
    __syncthreads();
    if (g == 0) store_rows(B.X, rows, d, tid);
    coop_linear(wf, rows, F, d, g, tid, red, [&](int m, int n, float v, int) { B.ACT[(long)m * F + n] = (bf16)gelu_(v); });
    wprefetch(wf, Ly.out, d, F, g, tid);
    lnprefetch(lf, Ly.ln_out, d, tid);
    grid_sync(bar, 24);
    // S5: y = act . W2^T + b2 + x'
    resfetch(rf, B.X, d, d, g, tid);
    load_rows(rows, B.ACT, F, tid);
    __syncthreads();
    coop_linear(wf, rows, d, F, g, tid, red, [&](int m, int n, float v, int o) { B.Y[(long)m * d + n] = (bf16)(v + rf.v[o]); });
    if (li + 1 < a.nl) {
      if constexpr (merged) qprefetch(qf, s_layers[li + 1].qkv, d, a.H, g, tid, 0);
      else wprefetch(wf, s_layers[li + 1].qkv, 3 * d, d, g, tid);
    } else {
      wprefetch(wf, a.transform, d, d, g, tid);
    }
    grid_sync(bar, 25);
    load_rows(rows, B.Y, d, tid);
    __syncthreads();
    CSTAMP(bar, 15);
  }
  // head: GELU(transform(LN(y)))
  ln_rows(rows, d, lf, tid);
  __syncthreads();
  coop_linear(wf, rows, d, d, g, tid, red, [&](int m, int n, float v, int) {
    if (m < nb) a.out_hidden[(long)m * d + n] = (bf16)gelu_(v);
  });
}

// ---- the Transformer decoder stack ------------------------------------------------------------------------------------------------
template <bool MERGED>
__global__ __launch_bounds__(CT) void dec_step_coop_kernel(const DecCoopArgs a) {
  __shared__ __attribute__((aligned(16))) bf16 rows[16 * C_LD];
  __shared__ float red[4096];
  __shared__ __attribute__((aligned(16))) float wscr[CW * 192];
  __shared__ unsigned s_base;
  const int tid = threadIdx.x, g = blockIdx.x;
  const int d = a.d, nb = a.nb, F = a.F, H = a.H, dk = a.d / a.H, pos = *a.pos;
  const CoopBufs& B = a.B;
  Bar bar;
  bar_init(bar, B.counter, B.err, &s_base);
  WFrag wf;
  LnFragT<MERGED ? 4 : C_MAXD / 64> lf;   // (the merged stage implies d <= 256)
  ResFrag rf;
  __shared__ emoasr_decoder_layer_t s_layers[8];   // (see lm_step_coop_kernel)
  __shared__ const void* s_kv[8];
  for (int i = tid; i < (int)(sizeof(emoasr_decoder_layer_t) / 4) * a.nl; i += CT)
    reinterpret_cast<unsigned*>(s_layers)[i] = reinterpret_cast<const unsigned*>(a.layers)[i];
  if (tid < a.nl) s_kv[tid] = a.kv[tid];
  __syncthreads();
  constexpr bool merged = MERGED;   // projection + self-attention in one stage (the host checks qkv_merge_ok)
  QFrag qf;
  if constexpr (merged) qprefetch(qf, s_layers[0].qkv, d, H, g, tid, 0);
  else wprefetch(wf, s_layers[0].qkv, 3 * d, d, g, tid);
  lnprefetch(lf, s_layers[0].ln1, d, tid);
  // the residual stream X lives in global memory; every workgroup builds the embedded rows for itself, workgroup 0 publishes them
  for (int i = tid; i < 16 * d; i += CT) {
    const int m = i / d, c = i - m * d;
    rows[m * C_LD + c] = m < nb ? (bf16)((float)a.embed[(long)a.ids[m] * d + c] * a.emb_scale + a.pe[(long)pos * d + c]) : (bf16)0.f;
  }
  __syncthreads();
  if (g == 0) store_rows(B.X, rows, d, tid);
  const long layer_elems = (long)nb * a.Lmax * d;
  const int nchunk = (a.T + 63) / 64, PS = dk + 2;
  for (int li = 0; li < a.nl; ++li) {
    const emoasr_decoder_layer_t& Ly = s_layers[li];
    bf16* kc = a.kcache + li * layer_elems;
    bf16* vc = a.vcache + li * layer_elems;
    // S1: h = LN1(x); q | k | v
    if constexpr (merged) qprefetch(qf, Ly.qkv, d, H, g, tid, 1);
    ln_rows(rows, d, lf, tid);
    __syncthreads();
    if constexpr (merged) {
      // S1 + S2: the head's q | k | v and its masked self-attention over the cached prefix, no barrier in between
      qkv_attn_stage(qf, rows, reinterpret_cast<bf16*>(red), nb, d, H, a.Lmax, kc, vc, pos, B.O, g, tid, wscr);
      wprefetch(wf, Ly.out, d, d, g, tid);
    } else {
      coop_linear(wf, rows, 3 * d, d, g, tid, red, [&](int m, int n, float v, int) {
        if (n < d) B.QKV[(long)m * 3 * d + n] = (bf16)v;
        else if (m < nb) {
          if (n < 2 * d) kc[((long)m * a.Lmax + pos) * d + n - d] = (bf16)v;
          else vc[((long)m * a.Lmax + pos) * d + n - 2 * d] = (bf16)v;
        }
      });
      wprefetch(wf, Ly.out, d, d, g, tid);
      grid_sync(bar);
      // S2: masked self-attention over the cached prefix
      coop_self_attention(B.QKV, nb, d, H, a.Lmax, kc, vc, pos, B.O, g, tid, wscr, bar);
    }
    lnprefetch(lf, Ly.ln2, d, tid);
    grid_sync(bar);
    // S3: x += o . Wout^T + b   (each element of X is read and written by the same thread)
    resfetch(rf, B.X, d, d, g, tid);
    load_rows(rows, B.O, d, tid);
    __syncthreads();
    coop_linear(wf, rows, d, d, g, tid, red, [&](int m, int n, float v, int o) { B.X[(long)m * d + n] = (bf16)(v + rf.v[o]); });
    wprefetch(wf, Ly.q2, d, d, g, tid);
    grid_sync(bar);
    // S4: q2 = LN2(x) . Wq2^T + b
    load_rows(rows, B.X, d, tid);
    __syncthreads();
    ln_rows(rows, d, lf, tid);
    __syncthreads();
    coop_linear(wf, rows, d, d, g, tid, red, [&](int m, int n, float v, int) { B.Q2[(long)m * d + n] = (bf16)v; });
    wprefetch(wf, Ly.out2, d, d, g, tid);
    lnprefetch(lf, Ly.ln3, d, tid);
    grid_sync(bar);
    // S5: source attention, (pair, 64-key chunk) items over all waves of all workgroups
    {
      int t5 = tid;
      EMO_FRESH(t5);
      const int wave = t5 >> 6, lane = t5 & 63;
      float* q = wscr + wave * 192;
      float* prob = q + 128;
      const bf16* kvl = uniform_ptr(static_cast<const bf16*>(s_kv[li]));
      const int nitem = nb * H * nchunk;
      for (int it = g + CG * wave; it < nitem; it += CG * CW) {
        const int pair = it / nchunk, ch = it - pair * nchunk;
        const int b = pair / H, hh = pair - b * H;
        const int tlen = min(a.kmem[b], a.T), t0 = 64 * ch;
        float m = -INFINITY, l = 0.f, o0 = 0.f, o1 = 0.f;
        if (t0 < tlen) {
          for (int c = lane; c < dk; c += 64) q[c] = (float)B.Q2[(long)b * d + hh * dk + c];
          const bf16* kb = kvl + ((long)b * a.T + t0) * 2 * d + hh * dk;
          attend64(q, dk, kb, 2 * d, kb + d, 2 * d, min(64, tlen - t0), 1.f / sqrtf((float)dk), prob, lane, m, l, o0, o1);
        }
        float* P = B.PART + ((long)pair * nchunk + ch) * PS;
        if (lane == 0) { P[0] = m; P[1] = l; }
        if (2 * lane < dk) { P[2 + 2 * lane] = o0; P[2 + 2 * lane + 1] = o1; }
      }
    }
    grid_sync(bar);
    // S6: merge the chunks into o2 while loading it; x += o2 . Wout2^T + b
    {
      resfetch(rf, B.X, d, d, g, tid);
      int t6 = tid;
      EMO_FRESH(t6);
      // (1) every chunk's (max, sum) -> LDS, one thread per (pair, chunk); (2) per pair: weights exp(m_c - max) / sum; (3) one
      // thread per output element: the weighted sum of the chunks' outputs, 8 loads in flight at a time
      const int npc = nb * H * nchunk;
      float* sm = red;            // [pairs * chunks] (<= 2048 floats each: `red` is idle here)
      float* sl = red + 2048;
      for (int i = t6; i < npc; i += CT) { sm[i] = B.PART[(long)i * PS]; sl[i] = B.PART[(long)i * PS + 1]; }
      __syncthreads();
      if (t6 < nb * H) {
        float mx = -INFINITY, l = 0.f;
        for (int ch = 0; ch < nchunk; ++ch) mx = fmaxf(mx, sm[t6 * nchunk + ch]);
        for (int ch = 0; ch < nchunk; ++ch) l += __expf(sm[t6 * nchunk + ch] - mx) * sl[t6 * nchunk + ch];
        const float inv = 1.f / l;
        for (int ch = 0; ch < nchunk; ++ch) sm[t6 * nchunk + ch] = __expf(sm[t6 * nchunk + ch] - mx) * inv;   // 0 past the memory's end
      }
      __syncthreads();
      const __amdgpu_buffer_rsrc_t rsP = make_rsrc(B.PART);
      for (int i = t6; i < 16 * d; i += CT) {
        const int m = i / d, c = i - m * d;
        float val = 0.f;
        if (m < nb) {
          const int pair = m * H + c / dk, dim = c % dk;
          for (int c0 = 0; c0 < nchunk; c0 += 8) {
            float pv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
              pv[e] = buf_load_f32<float>(rsP, c0 + e < nchunk ? (unsigned)((((long)pair * nchunk + c0 + e) * PS + 2 + dim) * 4) : EMO_OOB);
#pragma unroll
            for (int e = 0; e < 8; ++e) val += (c0 + e < nchunk ? sm[pair * nchunk + c0 + e] : 0.f) * pv[e];
          }
        }
        rows[m * C_LD + c] = (bf16)val;
      }
      __syncthreads();
      coop_linear(wf, rows, d, d, g, tid, red, [&](int m, int n, float v, int o) { B.X[(long)m * d + n] = (bf16)(v + rf.v[o]); });
    }
    wprefetch(wf, Ly.w1, F, d, g, tid);
    grid_sync(bar);
    // S7: act = ReLU(LN3(x) . W1^T + b1)
    load_rows(rows, B.X, d, tid);
    __syncthreads();
    ln_rows(rows, d, lf, tid);
    __syncthreads();
    coop_linear(wf, rows, F, d, g, tid, red, [&](int m, int n, float v, int) { B.ACT[(long)m * F + n] = (bf16)fmaxf(v, 0.f); });
    wprefetch(wf, Ly.w2, d, F, g, tid);
    if (li + 1 < a.nl) lnprefetch(lf, s_layers[li + 1].ln1, d, tid);
    grid_sync(bar);
    // S8: x += act . W2^T + b2
    resfetch(rf, B.X, d, d, g, tid);
    load_rows(rows, B.ACT, F, tid);
    __syncthreads();
    coop_linear(wf, rows, d, F, g, tid, red, [&](int m, int n, float v, int o) {
      const bf16 ov = (bf16)(v + rf.v[o]);
      B.X[(long)m * d + n] = ov;
      if (li + 1 == a.nl && m < nb) a.out_x[(long)m * d + n] = ov;
    });
    if (li + 1 < a.nl) {
      if constexpr (merged) qprefetch(qf, s_layers[li + 1].qkv, d, H, g, tid, 0);
      else wprefetch(wf, s_layers[li + 1].qkv, 3 * d, d, g, tid);
      grid_sync(bar);
      load_rows(rows, B.X, d, tid);
      __syncthreads();
    }
  }
}

int g_decode_coop = 1;
int g_coop_merge = 1;   // projection + self-attention in one stage where the shapes allow: bit 0 the decoder stack (339 -> 322 us),
                        // bit 1 the LM stack (351 -> 351 us: the later arrival of 15 of the 16 weight fragments and the four-fold
                        // projection eat the barrier it saves; off) -- option "decode_coop_merge"

// device scratch of the two chains (0: decoder, 1: LM), allocated on first use -- which must not be inside a stream capture: the
// search's eager warm-up pass (modeling/beam_search_device.py) comes first
CoopBufs* coop_bufs(int chain) {
  static CoopBufs bufs[2];
  static bool ready[2] = {false, false};
  if (!ready[chain]) {
    const size_t nrow = 16;
    const size_t bytes = 256 /* counter + err */ + nrow * 2 * (3 * C_MAXD + 3 * C_MAXD + C_MAXF + C_MAXD) +
                         (size_t)C_MAXPAIR * C_MAXCHUNK * 130 * 4;
    char* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess || hipMemset(p, 0, bytes) != hipSuccess) return nullptr;
    CoopBufs& b = bufs[chain];
    b.counter = reinterpret_cast<unsigned*>(p);
    b.err = reinterpret_cast<int*>(p + 64);
    bf16* q = reinterpret_cast<bf16*>(p + 256);
    b.X = q; q += nrow * C_MAXD;
    b.Y = q; q += nrow * C_MAXD;
    b.O = q; q += nrow * C_MAXD;
    b.QKV = q; q += nrow * 3 * C_MAXD;
    b.ACT = q; q += nrow * C_MAXF;
    b.Q2 = q; q += nrow * C_MAXD;
    b.PART = reinterpret_cast<float*>(q);
    ready[chain] = true;
  }
  return &bufs[chain];
}

// every linear of the stack must fit the (strip, k step) plan: at most 2 units per wave, k steps paired inside a strip
bool plan_ok(int N, int K) {
  if (K % 128 != 0 || N % 16 != 0) return false;   // (runs of 4 k steps must not straddle strips)
  const int ns = ((N + 15) / 16 + CG - 1) / CG, total = ns * (K / 32);
  return total <= C_UPW * CW && ns * 256 <= C_OPT * CT;
}

}  // namespace

static long g_coop_launches[2] = {0, 0};
void emo_decode_set_coop(int v) { g_decode_coop = v; }
void emo_decode_set_coop_merge(int v) { g_coop_merge = v; }

// The kernels are launched with <<<>>> (they are replayed from HIP graphs), so the co-residency of the CG workgroups their grid
// barrier assumes is verified once here: occupancy x compute units >= CG for all four instantiations.
static bool coop_resident() {
  static int cached = -1;
  if (cached < 0) {
    int dev = 0, cus = 0;
    bool ok = hipGetDevice(&dev) == hipSuccess &&
              hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess;
    const void* ks[4] = {(const void*)lm_step_coop_kernel<true>, (const void*)lm_step_coop_kernel<false>,
                         (const void*)dec_step_coop_kernel<true>, (const void*)dec_step_coop_kernel<false>};
    for (int i = 0; ok && i < 4; ++i) {
      int per_cu = 0;
      ok = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ks[i], CT, 0) == hipSuccess && (long)per_cu * cus >= CG;
    }
    cached = ok ? 1 : 0;
  }
  return cached == 1;
}

// Can the cooperative kernels take this step?  (bf16, <= 16 hypotheses, <= 12 / 8 layers, shapes that fit the plans)
bool emo_decode_coop_ok(int dtype, int nb, int nl, int max_layers, int d, int H, int F, int T) {
  return g_decode_coop && coop_resident() && dtype == EMO_BF16 && nb >= 1 && nb <= 16 && nl >= 1 && nl <= max_layers && d <= C_MAXD && F <= C_MAXF &&
         d % H == 0 && (d / H) % 8 == 0 && d / H <= 64 && nb * H <= C_MAXPAIR && (T + 63) / 64 <= C_MAXCHUNK && plan_ok(3 * d, d) &&
         plan_ok(d, d) && plan_ok(F, d) && plan_ok(d, F);
}

// The LM stack up to GELU(transform(x)) -> out_hidden [nb, d] (bf16); the caller applies ln_transform + the tied projection.
int emo_bert_lm_step_coop(int nl, const emoasr_bert_layer_t* layers, const emoasr_bert_step_t* io, void* out_hidden, hipStream_t s) {
  CoopBufs* B = coop_bufs(1);
  EMO_CHECK(B, "decode_coop: scratch allocation failed");
  LmCoopArgs a{};
  a.nl = nl; a.nb = io->nb; a.Lmax = io->Lmax; a.d = io->d; a.H = io->H; a.F = io->F;
  a.ids = io->ids; a.pos = io->pos; a.word_emb = (const bf16*)io->word_emb; a.pe = io->pe; a.ln_emb = io->ln_emb;
  a.kcache = (bf16*)io->kcache; a.vcache = (bf16*)io->vcache; a.transform = io->transform; a.out_hidden = (bf16*)out_hidden;
  a.B = *B;
  for (int i = 0; i < nl; ++i) a.layers[i] = layers[i];
  if ((g_coop_merge & 2) && qkv_merge_ok(a.d, a.H)) lm_step_coop_kernel<true><<<CG, CT, 0, s>>>(a);
  else lm_step_coop_kernel<false><<<CG, CT, 0, s>>>(a);
  EMO_LAUNCH_CHECK();
  ++g_coop_launches[1];
#ifdef EMO_COOP_STAMP
  {  // debug builds: the first (eager) launch's stamps of workgroup 0: mean ticks from the previous stamp, by stamp id
    static int calls = 0;
    if (calls++ == 0) {
      hipStreamSynchronize(s);
      static unsigned long long h[2][8192];
      int n[2] = {0, 0};
      hipMemcpyFromSymbol(n, HIP_SYMBOL(g_coop_nstamp), 8);
      hipMemcpyFromSymbol(h, HIP_SYMBOL(g_coop_stamps), sizeof(h));
      double acc[32] = {0}; int cnt[32] = {0};
      for (int k = 1; k < n[1]; ++k) {
        const int id = (int)(h[1][k] & 255);
        acc[id] += (double)((h[1][k] >> 8) - (h[1][k - 1] >> 8)); cnt[id]++;
      }
      fprintf(stderr, "[coop stamp] LM, %d stamps; mean ticks before each stamp id:", n[1]);
      for (int id = 0; id < 32; ++id) if (cnt[id]) fprintf(stderr, "  %d: %.0f (x%d)", id, acc[id] / cnt[id], cnt[id]);
      fprintf(stderr, "\n");
    }
  }
#endif
  return 0;
}

// The decoder stack up to (not including) the final LayerNorm -> out_x [nb, dd] (bf16).
int emo_transformer_decoder_step_coop(int nl, const emoasr_decoder_layer_t* layers, const emoasr_decoder_step_t* io, void* out_x,
                                      hipStream_t s) {
  CoopBufs* B = coop_bufs(0);
  EMO_CHECK(B, "decode_coop: scratch allocation failed");
  DecCoopArgs a{};
  a.nl = nl; a.nb = io->nb; a.Lmax = io->Lmax; a.T = io->T; a.d = io->dd; a.H = io->H; a.F = io->F;
  a.ids = io->ids; a.pos = io->pos; a.embed = (const bf16*)io->embed; a.pe = io->pe; a.emb_scale = io->emb_scale;
  a.kcache = (bf16*)io->kcache; a.vcache = (bf16*)io->vcache; a.kmem = io->kmem; a.out_x = (bf16*)out_x;
  a.B = *B;
  for (int i = 0; i < nl; ++i) { a.layers[i] = layers[i]; a.kv[i] = io->kv[i]; }
  if ((g_coop_merge & 1) && qkv_merge_ok(a.d, a.H)) dec_step_coop_kernel<true><<<CG, CT, 0, s>>>(a);
  else dec_step_coop_kernel<false><<<CG, CT, 0, s>>>(a);
  EMO_LAUNCH_CHECK();
  ++g_coop_launches[0];
  return 0;
}

// how many times the cooperative kernels were enqueued or captured into a graph (chain 0 decoder, 1 LM) since the library was
// loaded: lets a test assert that a search really took this path
extern "C" long emoasr_decode_coop_launches(int chain) { return chain == 0 || chain == 1 ? g_coop_launches[chain] : -1; }

// error flag of the barriers (a wait that gave up): 0 = fine.  Synchronises the device.
extern "C" long emoasr_decode_coop_status(void) {
  long worst = 0;
  for (int c = 0; c < 2; ++c) {
    CoopBufs* B = coop_bufs(c);
    if (!B) return -1;
    int e = 0;
    if (hipMemcpy(&e, B->err, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (e) worst = e;
  }
  return worst;
}
