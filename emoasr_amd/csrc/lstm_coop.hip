// The prediction network's LSTM recurrence (config 5, bf16) as ONE cooperative launch per layer and direction instead of two
// launches (recurrent GEMM + cell) per label position: H / 16 co-resident workgroups, each the owner of 16 hidden units for the whole
// sequence -- its 64 rows of W_hh (the four gates of its units) stay in LDS, its cell states in registers -- meet at a grid barrier
// once per position (the scheme of csrc/decode_coop.hip).
//
// Why: at the standard batch (36 utterances, ~80 label positions, H 512, 2 layers) the recurrence is ~640 dependent launches of
// 3-14 us per step (profiles/r02_l4_kernel_stats.csv: recurrent GEMMs 1.17 + 2.18 ms, cells 0.30 + 0.47 ms of a 17.6 ms step),
// each a [36 x 512] x [512 x 2048] product that no tiling can make efficient; per position the chip needs one barrier, one read of
// h_{u-1} (36 KB per workgroup) and 24-32 MFMAs per wave.
//
// MEASURED (MI355X, B 36, H 512, U ~100, rocprofv3): forward 326 us per layer (3.3 us per position: the 32-workgroup barrier is
// 2.8 of them), backward 516 us per layer (5.2 us per position); per training step 0.65 + 1.03 ms against 1.48 + 2.65 ms for the
// launch chain; config 5 1.69 -> 1.81 M frames/s (profiles/r02_l4c_kernel_stats.csv).
//
//   forward, position u : h_{u-1} [B][H] -> LDS; gates[m][4][16 own units] = h_{u-1} . W_own^T (16x16x32 MFMAs: a wave takes one
//                         gate and two 16-row tiles) + pre[u] (the input projection, requested before the barrier); the cell on
//                         one thread per (row, unit) pair, c in registers; h_u, c_u and the activated gates written out
//   backward, position u: cell backward for the own units (dc in registers); dgates -> global (the weight-gradient GEMMs run over
//                         the whole sequence afterwards) and LDS; partial dh_{u-1}[m][all H] = dgates_own [B x 64] . W_own [64 x H]
//                         as f32 per workgroup; after the barrier every workgroup sums the H / 16 partials of its own 16 columns
//                         in workgroup order (bit-reproducible; no float atomics)
//
// Numerics: the gate pre-activations are rounded to bf16 before the cell and dh_{u-1} is formed in f32 from the partials, as the
// launch chain's GEMM outputs were / were not; the k order of the products differs from the GEMM kernels', so results agree with
// the chain to bf16 rounding, not bit for bit (tests/test_ops_gpu.py compares the two paths; the config-5 goldens and the
// full-size parity tests run through this one).
//
// Reference: decoders/rnn_transducer.py:96-135 (nn.LSTM prediction network), torch LSTM gate order i, f, g, o.
#include <math.h>
#include <algorithm>
#include <mutex>
#include <vector>
#include "common.h"
#include "../../include/emoasr_hip.h"

namespace {

constexpr int LT = 512;          // threads per workgroup (8 waves)
constexpr int LW = LT / 64;
constexpr int L_MAXB = 64;       // rows (sequences) per batch GROUP: four 16-row MFMA tiles
constexpr int L_MAXH = 512;
// More than 64 sequences (the prediction network of several stacked micro-batches in one pass): ceil(B / 64) independent groups of
// H / 16 workgroups in ONE launch, each group with its own barrier counter -- the recurrence is bound by the per-position barrier
// and the dependent round trip, so 8 groups take the time of one.
constexpr int L_MAXGRP = 8;

#define EMO_FRESH(x) asm volatile("" : "+v"(x))

// The barrier's base (the counter value when the launch starts) is a KERNEL ARGUMENT kept by the host (lstm_base below): the
// counter is shared by every layer / model / hidden size of the process and G = H / 16 varies (2 .. 32, not always a power of
// two), so it cannot be recovered on the device as "the counter rounded down to a multiple of G".  Every launch performs exactly
// U - 1 whole barriers -- a workgroup whose wait gave up has still added its arrival -- so the host's running sum stays exact.
struct LBar { unsigned* counter; int* err; unsigned base, n, G; };
__device__ void lbar_init(LBar& b, unsigned* counter, int* err, unsigned G, unsigned base) {
  b.counter = counter; b.err = err; b.base = base; b.n = 0; b.G = G;
}
// a barrier of this launch (or an earlier one that nobody has reported yet) gave up: the outputs cannot be trusted
__device__ bool lbar_failed(const LBar& b) { return __hip_atomic_load(b.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; }
__device__ void lgrid_sync(LBar& b) {
  ++b.n;
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(b.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = b.base + b.G * b.n;
    unsigned spins = 0;
    while ((int)(__hip_atomic_load(b.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22)) { *b.err = 1; break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

__device__ __forceinline__ float rbf(float x) { return (float)(bf16)x; }   // round to bf16 and back

struct LstmFwdArgs {
  int U, B, H;
  const bf16* pre;     // [U][B][4H] input projection + biases
  const bf16* w_hh;    // [4H][H]
  const bf16* h0;      // [B][H] or null
  const float* c0;     // [B][H] or null
  bf16* hseq;          // [U][B][H]
  float* cseq;         // [U][B][H]
  bf16* gact;          // [U][B][4H] activated gates
  unsigned* counter; int* err; unsigned base[L_MAXGRP];
  int G;               // workgroups per batch group (H / 16); gridDim.x = G * ceil(B / 64)
};

// LDS: Ws [64][H + 8] | hs [64][H + 8] | gat [4][64][17] f32
__global__ __launch_bounds__(LT) void lstm_seq_fwd_kernel(const LstmFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int H = a.H, ld = H + 8, G = a.G, gi = blockIdx.x / G, g = blockIdx.x - gi * G;
  const int m0 = gi * L_MAXB, Bs = a.B, B = min(L_MAXB, Bs - m0);   // this group's rows m0 .. m0 + B of the Bs sequences
  bf16* Ws = reinterpret_cast<bf16*>(smem);
  bf16* hs = Ws + 64 * ld;
  float* gat = reinterpret_cast<float*>(hs + 64 * ld);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  LBar bar;
  lbar_init(bar, a.counter + 16 * gi, a.err, (unsigned)G, a.base[gi]);
  // the four gates' rows of the own 16 units: Ws[gate * 16 + j] = w_hh[gate * H + 16 g + j]
  for (int i = tid; i < 64 * (H / 8); i += LT) {
    const int r = i / (H / 8), c = (i - r * (H / 8)) * 8;
    *reinterpret_cast<bf16x8*>(Ws + r * ld + c) = *reinterpret_cast<const bf16x8*>(a.w_hh + ((long)(r >> 4) * H + 16 * g + (r & 15)) * H + c);
  }
  for (int i = tid; i < 64 * ld / 2; i += LT) reinterpret_cast<unsigned*>(hs)[i] = 0u;   // rows >= B stay zero
  __syncthreads();
  // this thread's two (row, unit) pairs: p = tid, tid + 512 -> m = p >> 4, n = p & 15
  float c_reg[2];
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    const int p = tid + LT * o, m = p >> 4, n = p & 15;
    c_reg[o] = (a.c0 && m < B) ? a.c0[(long)(m0 + m) * H + 16 * g + n] : 0.f;
  }
  const int gate = wave & 3, mt0 = 2 * (wave >> 2);
  for (int u = 0; u < a.U; ++u) {
    int t = tid;
    EMO_FRESH(t);
    // the input projection of this position for the own pairs (independent of the recurrence: before the wait)
    float pv[2][4];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int p = t + LT * o, m = p >> 4, n = p & 15;
#pragma unroll
      for (int q = 0; q < 4; ++q) pv[o][q] = m < B ? (float)a.pre[((long)u * Bs + m0 + m) * 4 * H + q * H + 16 * g + n] : 0.f;
    }
    const bf16* hprev = u > 0 ? a.hseq + ((long)(u - 1) * Bs + m0) * H : (a.h0 ? a.h0 + (long)m0 * H : nullptr);
    if (u > 0) lgrid_sync(bar);   // h_{u-1} complete on every workgroup
    if (hprev) {
      for (int i = t; i < B * (H / 8); i += LT) {
        const int m = i / (H / 8), c = (i - m * (H / 8)) * 8;
        *reinterpret_cast<bf16x8*>(hs + m * ld + c) = *reinterpret_cast<const bf16x8*>(hprev + (long)m * H + c);
      }
      __syncthreads();
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int mt = mt0 + mi;
        if (16 * mt < B) {
          f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
          const bf16* xrow = hs + (16 * mt + (lane & 15)) * ld + 8 * (lane >> 4);
          const bf16* wrow = Ws + (16 * gate + (lane & 15)) * ld + 8 * (lane >> 4);
          for (int ks = 0; ks < H / 32; ++ks)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(xrow + 32 * ks),
                                                          *reinterpret_cast<const bf16x8*>(wrow + 32 * ks), acc, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) gat[(gate * 64 + 16 * mt + 4 * (lane >> 4) + r) * 17 + (lane & 15)] = acc[r];
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int p = t + LT * o, m = p >> 4, n = p & 15;
      if (m < B) {
        float z[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) z[q] = rbf(pv[o][q] + (hprev ? gat[(q * 64 + m) * 17 + n] : 0.f));   // the chain's GEMM output was bf16
        const float ig = sigmoidf_(z[0]), fg = sigmoidf_(z[1]), gg = tanh_fast(z[2]), og = sigmoidf_(z[3]);
        const float cn = fg * c_reg[o] + ig * gg;
        c_reg[o] = cn;
        const long row = (long)u * Bs + m0 + m;
        a.cseq[row * H + 16 * g + n] = cn;
        a.hseq[row * H + 16 * g + n] = (bf16)(og * tanh_fast(cn));
        bf16* ga = a.gact + row * 4 * H + 16 * g + n;
        ga[0] = (bf16)ig; ga[H] = (bf16)fg; ga[2 * H] = (bf16)gg; ga[3 * H] = (bf16)og;
      }
    }
  }
  // a wait that gave up read a half-written h_{u-1}: poison the own slice of the outputs so that the loss / gradient norm turns
  // NaN and the optimizer's NaN skip drops the step (the host reports the flag at its next synchronisation point)
  if (lbar_failed(bar)) {
    const float qnan = __builtin_nanf("");
    for (long i = threadIdx.x; i < (long)a.U * B * 16; i += LT) {
      const long r = i >> 4, u = r / B, m = r - u * B;
      a.hseq[(u * Bs + m0 + m) * H + 16 * g + (i & 15)] = (bf16)qnan;
    }
  }
}

struct LstmBwdArgs {
  int U, B, H;
  const bf16* dh_seq;   // [U][B][H] gradient from above (dropout already applied)
  const bf16* gact;     // [U][B][4H]
  const float* cseq;    // [U][B][H]
  const float* c0;      // [B][H] or null
  const bf16* w_hh;     // [4H][H]
  bf16* dgp;            // [U][B][4H] out: gradient w.r.t. the gate pre-activations
  float* part;          // [groups][2][G][64][H] f32 scratch: every workgroup's contribution to dh_{u-1}
  unsigned* counter; int* err; unsigned base[L_MAXGRP];
  int G;
};

// LDS: WT [H][72] (the own 64 gate rows of W_hh, transposed: WT[n][k] = w_hh[row(k)][n]) | dg [64][72] (the own dgates of the position)
__global__ __launch_bounds__(LT) void lstm_seq_bwd_kernel(const LstmBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int H = a.H, G = a.G, gi = blockIdx.x / G, g = blockIdx.x - gi * G;
  const int m0 = gi * L_MAXB, Bs = a.B, B = min(L_MAXB, Bs - m0);
  constexpr int LDK = 72;
  bf16* WT = reinterpret_cast<bf16*>(smem);
  bf16* dg = WT + H * LDK;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  LBar bar;
  lbar_init(bar, a.counter + 16 * gi, a.err, (unsigned)G, a.base[gi]);
  for (int i = tid; i < 64 * H; i += LT) {   // k = gate * 16 + j <-> row gate * H + 16 g + j
    const int k = i / H, n = i - k * H;
    WT[n * LDK + k] = a.w_hh[((long)(k >> 4) * H + 16 * g + (k & 15)) * H + n];
  }
  for (int i = tid; i < 64 * LDK / 2; i += LT) reinterpret_cast<unsigned*>(dg)[i] = 0u;   // rows >= B stay zero
  __syncthreads();
  float dc_reg[2] = {0.f, 0.f};
  const long slab = (long)64 * H;   // one workgroup's partial
  float* const part = a.part + (long)gi * 2 * G * slab;
  for (int u = a.U - 1; u >= 0; --u) {
    int t = tid;
    EMO_FRESH(t);
    // everything of this position that does not depend on the recurrence: before the wait
    float dho[2], gv[2][4], cu[2], cp[2];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int p = t + LT * o, m = p >> 4, n = p & 15;
      const long row = (long)u * Bs + m0 + m, col = 16 * g + n;
      const bool ok = m < B;
      dho[o] = ok ? (float)a.dh_seq[row * H + col] : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) gv[o][q] = ok ? (float)a.gact[row * 4 * H + q * H + col] : 0.f;
      cu[o] = ok ? a.cseq[row * H + col] : 0.f;
      cp[o] = !ok ? 0.f : (u > 0 ? a.cseq[(row - Bs) * H + col] : (a.c0 ? a.c0[(long)(m0 + m) * H + col] : 0.f));
    }
    const bool rec = u < a.U - 1;   // position u + 1 contributed to dh_u
    if (rec) lgrid_sync(bar);
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int p = t + LT * o, m = p >> 4, n = p & 15;
      if (m < B) {
        float dh = dho[o];
        if (rec) {   // the workgroups' partials of the own column, in workgroup order
          const float* src = part + ((long)((u + 1) & 1) * G) * slab + (long)m * H + 16 * g + n;
          float acc = 0.f;
          for (int w0 = 0; w0 < G; w0 += 8) {
            float pv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) pv[e] = w0 + e < G ? src[(long)(w0 + e) * slab] : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += pv[e];
          }
          dh += acc;
        }
        const float ig = gv[o][0], fg = gv[o][1], gg = gv[o][2], og = gv[o][3];
        const float tc = tanh_fast(cu[o]);
        const float dct = dc_reg[o] + dh * og * (1.f - tc * tc);
        const bf16 d0 = (bf16)(dct * gg * ig * (1.f - ig)), d1 = (bf16)(dct * cp[o] * fg * (1.f - fg));
        const bf16 d2 = (bf16)(dct * ig * (1.f - gg * gg)), d3 = (bf16)(dh * tc * og * (1.f - og));
        dc_reg[o] = dct * fg;
        bf16* dst = a.dgp + ((long)u * Bs + m0 + m) * 4 * H + 16 * g + n;
        dst[0] = d0; dst[H] = d1; dst[2 * H] = d2; dst[3 * H] = d3;
        dg[m * LDK + n] = d0; dg[m * LDK + 16 + n] = d1; dg[m * LDK + 32 + n] = d2; dg[m * LDK + 48 + n] = d3;
      }
    }
    if (u > 0) {
      __syncthreads();
      // partial dh_{u-1}[m][:] = dg [B x 64] . W_own [64 x H]: column strips of 16 dealt to the waves
      float* dstp = part + ((long)(u & 1) * G + g) * slab;
      const int MT = (B + 15) / 16;
      for (int st = wave; st < H / 16; st += LW) {
        bf16x8 wf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wf[ks] = *reinterpret_cast<const bf16x8*>(WT + (16 * st + (lane & 15)) * LDK + 32 * ks + 8 * (lane >> 4));
        for (int mt = 0; mt < MT; ++mt) {
          f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(dg + (16 * mt + (lane & 15)) * LDK + 32 * ks + 8 * (lane >> 4)),
                                                          wf[ks], acc, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = 16 * mt + 4 * (lane >> 4) + r;
            if (m < B) dstp[(long)m * H + 16 * st + (lane & 15)] = acc[r];
          }
        }
      }
      __syncthreads();   // dg is rewritten by the next position
    }
  }
  if (lbar_failed(bar)) {   // (see the forward kernel)
    const float qnan = __builtin_nanf("");
    for (long i = threadIdx.x; i < (long)a.U * B * 64; i += LT) {
      const long r = i >> 6, u = r / B, m = r - u * B; const int k = (int)(i & 63);
      a.dgp[(u * Bs + m0 + m) * 4 * H + (long)(k >> 4) * H + 16 * g + (k & 15)] = (bf16)qnan;
    }
  }
}

constexpr int L_CNT_BYTES = 4096;   // [2 directions][L_MAXGRP groups] counters, 64 bytes apart; the error flag behind them
// The barrier counters and the host-tracked value each one holds at the start of the next launch belong to ONE stream: launches on
// a stream are ordered, so the host knows the counter's value without reading it back.  One area per (device, stream)
// (emo_stream_scratch: EmoScratch::host carries the expected values, [direction][group]); the launches refuse a capturing stream --
// a replayed graph would start from counter values the host no longer tracks.
static_assert(2 * L_MAXGRP <= 32, "expected counter values live in EmoScratch::host");
std::mutex g_lstm_mu;
std::vector<EmoScratch*> g_lstm_areas;   // every area handed out (emoasr_lstm_coop_status walks them)
EmoScratch* lstm_area(void* stream) {
  EmoScratch* sc = emo_stream_scratch(EMO_SCRATCH_LSTM, stream, L_CNT_BYTES);
  if (sc) {
    std::lock_guard<std::mutex> lock(g_lstm_mu);
    if (std::find(g_lstm_areas.begin(), g_lstm_areas.end(), sc) == g_lstm_areas.end()) g_lstm_areas.push_back(sc);
  }
  return sc;
}
// Cooperative launches of DIFFERENT streams must not overlap: the residency check below is per launch, and two launches that are each
// partly resident would spin on their barriers until the give-up flag fires (two large recurrences -- B > 64 groups at H = 512, one
// workgroup per CU -- on two streams).  A process-wide event chain per device orders them: a launch waits for the previous
// cooperative launch if that went to another stream, and records the event behind itself.  (Same stream: already ordered.)
struct CoopChain { hipEvent_t ev = nullptr; void* last = nullptr; bool armed = false; };
CoopChain g_coop_chain[16];
std::mutex g_coop_mu;
struct CoopOrder {
  CoopChain* ch = nullptr;
  void* stream;
  std::unique_lock<std::mutex> lock;
  explicit CoopOrder(void* s) : stream(s), lock(g_coop_mu) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return;
    ch = &g_coop_chain[dev];
    if (!ch->ev && hipEventCreateWithFlags(&ch->ev, hipEventDisableTiming) != hipSuccess) { ch->ev = nullptr; ch = nullptr; return; }
    if (ch->armed && ch->last != stream) hipStreamWaitEvent((hipStream_t)stream, ch->ev, 0);
  }
  ~CoopOrder() {   // (runs after the launch: the event sits right behind the cooperative kernel)
    if (ch && hipEventRecord(ch->ev, (hipStream_t)stream) == hipSuccess) { ch->armed = true; ch->last = stream; }
  }
};

unsigned* lstm_counter(EmoScratch* sc, int which, int** err) {
  unsigned* buf = static_cast<unsigned*>(sc->dev);
  *err = reinterpret_cast<int*>(buf) + 2 * L_MAXGRP * 16;
  return buf + 16 * L_MAXGRP * which;
}
bool lstm_capturing(void* stream) {
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone;
}

int g_lstm_coop = 1;

// counter value at the start of the next launch, per counter (0 forward, 1 backward); launches are stream-ordered
void lstm_base(EmoScratch* sc, int which, int G, int U, int ngrp, unsigned (&base)[L_MAXGRP]) {
  unsigned* tracked = sc->host + which * L_MAXGRP;
  for (int i = 0; i < L_MAXGRP; ++i) {
    base[i] = tracked[i];
    if (i < ngrp) tracked[i] += (unsigned)G * (unsigned)(U - 1);
  }
}
int lstm_groups(int B) { return (B + L_MAXB - 1) / L_MAXB; }

// The kernels are launched with <<<>>>, so the co-residency of their G workgroups is verified here instead (once per kernel and
// LDS size): the grid barrier would otherwise spin until its bail-out.
int lstm_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
  }
  return cus;
}
// workgroups of this kernel the device holds at once (0 = the query failed)
long lstm_capacity(const void* kernel, size_t smem) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, LT, smem) != hipSuccess) return 0;
  return (long)per_cu * lstm_cus();
}

}  // namespace

void emo_lstm_set_coop(int v) { g_lstm_coop = v; }

// dynamic LDS of the two kernels, and the workgroups of each the device holds at once (attribute set + occupancy query, cached per
// size; 0 = the query failed).  emoasr_lstm_seq_supported and the two launches share these, so "supported" and the launch-time
// residency check cannot disagree.
static size_t lstm_fwd_smem(int H) { return (size_t)2 * 64 * (H + 8) * 2 + 4 * 64 * 17 * 4; }
static size_t lstm_bwd_smem(int H) { return (size_t)(H + 64) * 72 * 2; }
static long lstm_cap_of(int which, int H) {
  static size_t set_bytes[2] = {0, 0}, cap_bytes[2] = {0, 0};
  static long cap[2] = {0, 0};
  const void* k = which ? (const void*)lstm_seq_bwd_kernel : (const void*)lstm_seq_fwd_kernel;
  const size_t smem = which ? lstm_bwd_smem(H) : lstm_fwd_smem(H);
  if (smem > set_bytes[which]) {
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return 0;
    set_bytes[which] = smem;
  }
  if (smem != cap_bytes[which]) { cap[which] = lstm_capacity(k, smem); cap_bytes[which] = smem; }
  return cap[which];
}

// Can the cooperative recurrence take this layer?  (bf16, B <= 8 groups of 64, H a multiple of 32 up to 512, and every group's
// workgroups of BOTH kernels resident together by the occupancy query -- the same numbers the launches check)
extern "C" int emoasr_lstm_seq_supported(int dtype, int B, int H) {
  if (!(g_lstm_coop && dtype == EMO_BF16 && B >= 1 && B <= L_MAXB * L_MAXGRP && H % 32 == 0 && H >= 32 && H <= L_MAXH)) return 0;
  const long need = (long)lstm_groups(B) * (H / 16);
  return lstm_cap_of(0, H) >= need && lstm_cap_of(1, H) >= need;
}

// hseq[u], cseq[u], gact[u] for u = 0 .. U - 1 from pre[u] = x_u . W_ih^T + b (bf16 [U][B][4H]) and the recurrent weights
extern "C" int emoasr_lstm_seq_fwd(int dtype, int U, int B, int H, const void* pre, const void* w_hh, const void* h0, const float* c0,
                                   void* hseq, float* cseq, void* gact, void* stream) {
  EMO_CHECK(emoasr_lstm_seq_supported(dtype, B, H), "lstm_seq_fwd: unsupported shape (dtype %d, B %d, H %d)", dtype, B, H);
  if (U == 0) return 0;
  LstmFwdArgs a{};
  a.U = U; a.B = B; a.H = H;
  a.pre = (const bf16*)pre; a.w_hh = (const bf16*)w_hh; a.h0 = (const bf16*)h0; a.c0 = c0;
  a.hseq = (bf16*)hseq; a.cseq = cseq; a.gact = (bf16*)gact;
  EMO_CHECK(!lstm_capturing(stream), "lstm_seq_fwd: the cooperative recurrence cannot be captured into a graph (host-tracked barrier values)");
  EmoScratch* sc = lstm_area(stream);
  if (!sc) return 1;
  a.counter = lstm_counter(sc, 0, &a.err);
  const size_t smem = lstm_fwd_smem(H);
  const long cap = lstm_cap_of(0, H);
  const int ngrp = lstm_groups(B);
  a.G = H / 16;
  EMO_CHECK(cap >= (long)a.G * ngrp, "lstm_seq_fwd: the device holds %ld of the %d workgroups at once", cap, a.G * ngrp);
  lstm_base(sc, 0, a.G, U, ngrp, a.base);
  {
    CoopOrder order(stream);
    lstm_seq_fwd_kernel<<<a.G * ngrp, LT, smem, (hipStream_t)stream>>>(a);
  }
  EMO_LAUNCH_CHECK();
  return 0;
}

// scratch of emoasr_lstm_seq_bwd: [ceil(B / 64)][2][H / 16][64][H] f32
extern "C" long emoasr_lstm_seq_bwd_ws_bytes(int B, int H) { return (long)lstm_groups(B) * 2 * (H / 16) * 64 * H * 4; }

// dgp[u] (gradient w.r.t. the gate pre-activations, [U][B][4H]) for u = U - 1 .. 0 from dh_seq (gradient w.r.t. the layer's
// outputs), the stored activated gates and cell states; the weight / input gradients follow from dgp as before
extern "C" int emoasr_lstm_seq_bwd(int dtype, int U, int B, int H, const void* dh_seq, const void* gact, const float* cseq,
                                   const float* c0, const void* w_hh, void* dgp, void* ws, long ws_bytes, void* stream) {
  EMO_CHECK(emoasr_lstm_seq_supported(dtype, B, H), "lstm_seq_bwd: unsupported shape (dtype %d, B %d, H %d)", dtype, B, H);
  EMO_CHECK(ws && ws_bytes >= emoasr_lstm_seq_bwd_ws_bytes(B, H), "lstm_seq_bwd: scratch too small");
  if (U == 0) return 0;
  LstmBwdArgs a{};
  a.U = U; a.B = B; a.H = H;
  a.dh_seq = (const bf16*)dh_seq; a.gact = (const bf16*)gact; a.cseq = cseq; a.c0 = c0; a.w_hh = (const bf16*)w_hh;
  a.dgp = (bf16*)dgp; a.part = (float*)ws;
  EMO_CHECK(!lstm_capturing(stream), "lstm_seq_bwd: the cooperative recurrence cannot be captured into a graph (host-tracked barrier values)");
  EmoScratch* sc = lstm_area(stream);
  if (!sc) return 1;
  a.counter = lstm_counter(sc, 1, &a.err);
  const size_t smem = lstm_bwd_smem(H);
  const long cap = lstm_cap_of(1, H);
  const int ngrp = lstm_groups(B);
  a.G = H / 16;
  EMO_CHECK(cap >= (long)a.G * ngrp, "lstm_seq_bwd: the device holds %ld of the %d workgroups at once", cap, a.G * ngrp);
  lstm_base(sc, 1, a.G, U, ngrp, a.base);
  {
    CoopOrder order(stream);
    lstm_seq_bwd_kernel<<<a.G * ngrp, LT, smem, (hipStream_t)stream>>>(a);
  }
  EMO_LAUNCH_CHECK();
  return 0;
}

// error flag of the recurrence kernels' barriers (a wait that gave up): 0 = fine.  Synchronises the device; a reported
// failure is cleared (flag and both counters back to zero), so the launches that follow start from a clean state.
extern "C" long emoasr_lstm_coop_status(void) {
  std::vector<EmoScratch*> areas;
  {
    std::lock_guard<std::mutex> lock(g_lstm_mu);
    areas = g_lstm_areas;
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  long any = 0;
  for (EmoScratch* sc : areas) {   // every stream's area of this process (areas of other devices answer through their own pointers)
    int* err = nullptr;
    unsigned* c = lstm_counter(sc, 0, &err);
    int e = 0;
    if (hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (e) {
      if (hipDeviceSynchronize() != hipSuccess || hipMemset(c, 0, L_CNT_BYTES) != hipSuccess) return -1;
      for (int i = 0; i < 2 * L_MAXGRP; ++i) sc->host[i] = 0u;
      any = e;
    }
  }
  return any;
}
