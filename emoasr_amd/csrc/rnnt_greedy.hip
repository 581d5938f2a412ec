// Time-synchronous greedy transducer search (asr/modeling/decoders/rnn_transducer.py:194-240) for ONE utterance as ONE launch:
//
//     dout, state = recurrency(<sos>);  t = 0
//     while t < T:  tok = argmax_v output(tanh(w_enc e_t + w_dec dout))      # one joint evaluation per step
//                   blank -> t += 1      else -> emit tok, dout, state = recurrency(tok, state)      stop after max_seq_len labels
//
// The loop is a chain of tiny dependent products (a [V x J] GEMV per step, two LSTM layers + a [J x H] GEMV per emitted label) whose
// control flow depends on their result.  Run from the host it costs one round trip and ~10 launches per emitted label
// (engine.rnnt_greedy: 110 us per label, RTF 2.7e-3 on the bench's utterances).  Here G co-resident workgroups run the whole
// search: every workgroup owns a slice of each matrix' rows (its slices of the output and w_dec matrices live in LDS for the whole
// launch, the LSTM rows stream from L2), the hidden state / joint input / per-workgroup arg-max candidates are exchanged through
// small global buffers as DATA-TAGGED words: a value and the step it belongs to travel in one 8-byte relaxed atomic store, readers
// poll the words they need until the tag is the expected one.  There is no grid barrier and no fence in the kernel: a dependent
// product starts as soon as its own inputs have arrived (first version, with csrc/decode_coop.hip's arrival-counter barrier
// between the phases: 4.6 us per phase -- store drain, L2 write-back, arrival, poll, fetch are five dependent round trips;
// tagged words: one store and one poll).  Every workgroup takes the same decisions from the same data, so no control
// information is exchanged at all.
//     blank step   : joint GEMV + arg-max candidates -> 1 hand-off
//     emitted label: LSTM layer 0 -> hand-off -> LSTM layer 1 -> hand-off -> w_dec -> hand-off (+ the joint step above)
//
// Numerics follow the launch chain it replaces: products accumulate in f32, every intermediate (gate pre-activations, h, the joint
// input, logits) is rounded to the compute dtype where the chain's kernels store it; the summation ORDER inside a dot product
// differs, so decisions can differ on near-ties of the arg-max (tests: identical ids on the fitted golden model and the full-size
// f32 parity test; bf16 is held to the same agreement as the chain).
#include <math.h>
#include "common.h"
#include "../../include/emoasr_hip.h"

namespace {

constexpr int GT = 256;            // threads per workgroup
constexpr int G_MAXH = 1024, G_MAXJ = 1024, G_MAXE = 1024;

// Data-tagged hand-off (no barrier, no fence): a value travels as ONE 8-byte word {payload, tag}, written with a relaxed
// device-scope atomic store and polled with relaxed device-scope atomic loads until the tag is the expected one.  A workgroup
// cannot finish phase p + 1 before every workgroup has produced phase p, so producers are never more than one phase ahead of any
// reader: a buffer rewritten every second phase or less often is safe with one copy per parity.
__device__ __forceinline__ void put64(unsigned long long* p, unsigned hi, unsigned lo) {
  __hip_atomic_store(p, ((unsigned long long)hi << 32) | lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// polls until (word & mask) == want; returns the word (sets *err and returns the last value after ~2^22 polls)
__device__ __forceinline__ unsigned long long get64(const unsigned long long* p, unsigned long long mask, unsigned long long want,
                                                    int* err) {
  unsigned long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned spins = 0;
  while ((v & mask) != want) {
    __builtin_amdgcn_s_sleep(1);
    v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (++spins > (1u << 22)) { *err = 1; break; }
  }
  return v;
}

template <typename T> __device__ __forceinline__ float rnd(float x) { return to_f32(from_f32<T>(x)); }   // round to T and back

// out[r] = sum_k W[r][k] * x[k] for r < nrows (f32 accumulation): 8 lanes per row, 16-byte loads, xor-shuffle reduction; the
// result is valid on every lane of the row's group.  W: row-major with row stride ldw (global or LDS), x: f32 in LDS.
template <typename T>
__device__ __forceinline__ float row_dot(const T* __restrict__ w, int K, const float* __restrict__ x, int sub) {
  constexpr int VEC = 16 / sizeof(T);
  float acc = 0.f;
  for (int k = sub * VEC; k < K; k += 8 * VEC) {
    float wv[VEC];
    if constexpr (sizeof(T) == 2) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(w + k);
#pragma unroll
      for (int e = 0; e < 8; ++e) wv[e] = (float)v[e];
    } else {
      const f32x4 v = *reinterpret_cast<const f32x4*>(w + k);
#pragma unroll
      for (int e = 0; e < 4; ++e) wv[e] = v[e];
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc += wv[e] * x[k + e];
  }
  acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64);
  return acc;
}

struct GreedyArgs {
  int T, E, H, J, V, blank, eos, max_len;
  const void* e_all;                 // [T][J] = w_enc . eouts + bias (compute dtype)
  const void* emb;                   // [V][E]
  const void *w_ih[2], *w_hh[2];     // [4H][in], [4H][H]
  const float* b_lstm[2];            // [4H] = bias_ih + bias_hh
  const void* w_dec; const float* b_dec;   // [J][H], [J]
  const void* w_out; const float* b_out;   // [V][J], [V]
  unsigned long long* hbuf;          // [2][H]  {h value bits, label-step tag} of the two layers
  unsigned long long* gbuf;          // [J]     {w_dec . dout + bias, label-step tag}
  unsigned long long* part;          // [2][G]  {candidate value bits, step tag (12 bits) | index (20 bits)}, by step parity
  int* hyp; int* align; int* lens;   // [max_len + 1], [T + max_len + 1], [2] = {len(hyp), len(align)}
  unsigned* counter; int* err;
};

// LDS (f32 vectors first): emb [E] | h0 [H] | h1 [H] | g [J] | e_t [J] | hj [J] | red [128] | own biases (LSTM 2 x 4 nu, output nv,
// w_dec nj) | own cell states [2 nu]; then the stationary weight slices (compute dtype, rows padded by 16 bytes).
// Every vector a phase needs is LDS-resident except the ONE that the previous phase produced on other workgroups: a phase costs one
// global round trip for that vector plus the barrier (store drain, arrival, poll) -- the round trips, not the arithmetic, are its time.
template <typename T, bool WLDS>
__global__ __launch_bounds__(GT) void rnnt_greedy_kernel(const GreedyArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int H = a.H, J = a.J, V = a.V, E = a.E;
  const int g = blockIdx.x, G = gridDim.x, tid = threadIdx.x, lane = tid & 63;
  const int grp = tid >> 3, sub = tid & 7;            // 32 row groups of 8 lanes
  const int v0 = (int)((long)V * g / G), v1 = (int)((long)V * (g + 1) / G), nv = v1 - v0;
  const int j0 = (int)((long)J * g / G), j1 = (int)((long)J * (g + 1) / G), nj = j1 - j0;
  const int u0 = (int)((long)H * g / G), u1 = (int)((long)H * (g + 1) / G), nu = u1 - u0;
  const int nvmax = (V + G - 1) / G + 1, njmax = (J + G - 1) / G + 1;
  float* embx = reinterpret_cast<float*>(smem);
  float* h0c = embx + E;
  float* h1c = h0c + H;
  float* gL = h1c + H;
  float* etL = gL + J;
  float* hj = etL + J;
  float* red = hj + J;                                // [128]
  float* bl = red + 128;                              // [2][4 nu]
  float* bo = bl + 8 * nu;                            // [nvmax]
  float* bd = bo + nvmax;                             // [njmax]
  float* cL = bd + njmax;                             // [2][nu]
  // the stationary slices are stored with 16 bytes of padding per row: the 8 row groups of a wave read the same columns of 8
  // different rows at once (a row stride of a multiple of 128 bytes would put them on the same banks)
  constexpr int PAD = 16 / sizeof(T);
  constexpr int VEC = 16 / sizeof(T);
  const int ldo = J + PAD, ldd = H + PAD;
  const int nfl = E + 2 * H + 3 * J + 128 + 8 * nu + nvmax + njmax + 2 * nu;
  T* wo_s = reinterpret_cast<T*>(smem + ((size_t)nfl * 4 + 15) / 16 * 16);   // this workgroup's rows of w_out: [nv][J + PAD]
  T* wd_s = wo_s + (long)nvmax * ldo;                                        // ... of w_dec: [nj][H + PAD]
  {
    const T* wo = static_cast<const T*>(a.w_out) + (long)v0 * J;
    for (int i = tid * VEC; i < nv * J; i += GT * VEC) {
      const int r = i / J, c = i - r * J;
      *reinterpret_cast<u32x4*>(wo_s + (long)r * ldo + c) = *reinterpret_cast<const u32x4*>(wo + i);
    }
    const T* wd = static_cast<const T*>(a.w_dec) + (long)j0 * H;
    for (int i = tid * VEC; i < nj * H; i += GT * VEC) {
      const int r = i / H, c = i - r * H;
      *reinterpret_cast<u32x4*>(wd_s + (long)r * ldd + c) = *reinterpret_cast<const u32x4*>(wd + i);
    }
    for (int i = tid; i < nv; i += GT) bo[i] = a.b_out[v0 + i];
    for (int i = tid; i < nj; i += GT) bd[i] = a.b_dec[j0 + i];
    for (int i = tid; i < 8 * nu; i += GT) {
      const int l = i / (4 * nu), r = i - l * 4 * nu;
      bl[i] = a.b_lstm[l][(long)(r / nu) * H + u0 + r % nu];
    }
    for (int i = tid; i < 2 * nu; i += GT) cL[i] = 0.f;
    for (int i = tid; i < H; i += GT) { h0c[i] = 0.f; h1c[i] = 0.f; }
  }
  // LDS-resident LSTM rows: layer l at wl_s[l], row r = q * nu + (u - u0) holds [w_ih row (nin) | pad | w_hh row (H) | pad]
  T* wl_s[2] = {nullptr, nullptr};
  int ldl[2] = {0, 0};
  if constexpr (WLDS) {
    T* base = wd_s + (long)njmax * ldd;
    for (int l = 0; l < 2; ++l) {
      const int nin = l == 0 ? E : H;
      ldl[l] = nin + PAD + H + PAD;
      wl_s[l] = base;
      base += (long)4 * nu * ldl[l];
      const T* wih = static_cast<const T*>(a.w_ih[l]);
      const T* whh = static_cast<const T*>(a.w_hh[l]);
      const int per_row = (nin + H) / VEC;
      for (int i = tid; i < 4 * nu * per_row; i += GT) {
        const int r = i / per_row, c = (i - r * per_row) * VEC;
        const long row = (long)(r / nu) * H + u0 + r % nu;
        if (c < nin) *reinterpret_cast<u32x4*>(wl_s[l] + (long)r * ldl[l] + c) = *reinterpret_cast<const u32x4*>(wih + row * nin + c);
        else *reinterpret_cast<u32x4*>(wl_s[l] + (long)r * ldl[l] + nin + PAD + (c - nin)) = *reinterpret_cast<const u32x4*>(whh + row * H + (c - nin));
      }
    }
  }
  __syncthreads();
  // exchange buffers (global, data-tagged words): h of layer l at hx + l * H, g at a.gbuf -- rewritten once per label (four
  // phases apart); the arg-max candidates once per step, by step parity
  unsigned long long* hx = a.hbuf;
  unsigned nlab = 0;   // label-step tag (1-based)
  auto fetch = [&](float* dst, const unsigned long long* src, int n, unsigned tag) {   // all n words of a vector, once tagged `tag`
    for (int i = tid; i < n; i += GT) dst[i] = __uint_as_float((unsigned)(get64(src + i, 0xFFFFFFFFull, tag, a.err) >> 32));
    __syncthreads();
  };

  // ---- prediction network for one token, then w_dec: leaves g (LDS) = w_dec . dout + bias (three hand-offs, no barrier) -----------------------------------------
  auto decoder_step = [&](const int tok) {
    ++nlab;
    {
      const T* er = static_cast<const T*>(a.emb) + (long)tok * E;
      for (int i = tid; i < E; i += GT) embx[i] = to_f32(er[i]);
      __syncthreads();
    }
    for (int l = 0; l < 2; ++l) {
      const int nin = l == 0 ? E : H;
      const float* xin = l == 0 ? embx : h0c;   // layer 1: the NEW hidden state of layer 0 (fetched below)
      float* hprev = l == 0 ? h0c : h1c;        // this layer's hidden state after the previous token
      const T* wih = static_cast<const T*>(a.w_ih[l]);
      const T* whh = static_cast<const T*>(a.w_hh[l]);
      // rows of the own units: unit u, gate q -> row q * H + u; 32 (unit, gate) rows per pass
      for (int r0 = 0; r0 < 4 * nu; r0 += 32) {
        const int r = r0 + grp;
        if (r < 4 * nu) {
          const long row = (long)(r / nu) * H + u0 + r % nu;
          // the chain: pre = round(W_ih x + b); gates = round(W_hh h + pre)   (h = 0 before the first token: gates = pre)
          float pre, rec;
          if constexpr (WLDS) {
            const T* wr = wl_s[l] + (long)r * ldl[l];
            pre = rnd<T>(row_dot<T>(wr, nin, xin, sub) + bl[l * 4 * nu + r]);
            rec = row_dot<T>(wr + nin + PAD, H, hprev, sub);
          } else {
            pre = rnd<T>(row_dot<T>(wih + row * nin, nin, xin, sub) + bl[l * 4 * nu + r]);
            rec = row_dot<T>(whh + row * H, H, hprev, sub);
          }
          if (sub == 0) red[r] = rnd<T>(rec + pre);
        }
      }
      __syncthreads();
      if (tid < nu) {
        const float ig = sigmoid_t<T>(red[tid]), fg = sigmoid_t<T>(red[nu + tid]), gg = tanh_t<T>(red[2 * nu + tid]), og = sigmoid_t<T>(red[3 * nu + tid]);
        const float cn = fg * cL[l * nu + tid] + ig * gg;
        cL[l * nu + tid] = cn;
        put64(hx + (long)l * H + u0 + tid, __float_as_uint(rnd<T>(og * tanh_t<T>(cn))), nlab);
      }
      __syncthreads();   // (red and hprev are rewritten below)
      fetch(hprev, hx + (long)l * H, H, nlab);  // the layer's new hidden state, all units (layer 1 reads it as its input next)
    }
    // g = round(w_dec . dout + bias): the own rows, from LDS
    for (int r0 = 0; r0 < nj; r0 += 32) {
      const int r = r0 + grp;
      if (r < nj) {
        const float v = rnd<T>(row_dot<T>(wd_s + (long)r * ldd, H, h1c, sub) + bd[r]);
        if (sub == 0) put64(a.gbuf + j0 + r, __float_as_uint(v), nlab);
      }
    }
    fetch(gL, a.gbuf, J, nlab);
  };

  decoder_step(a.eos);
  int t = 0, nh = 0, na = 0, step = 0;
  {
    const T* et = static_cast<const T*>(a.e_all);
    for (int i = tid; i < J; i += GT) etL[i] = to_f32(et[i]);
    __syncthreads();
  }
  while (t < a.T) {
    // the next frame's encoder projection is requested now and committed only if this step turns out blank
    float enext[4];
    {
      const T* en = static_cast<const T*>(a.e_all) + (long)min(t + 1, a.T - 1) * J;
#pragma unroll
      for (int k = 0; k < 4; ++k) enext[k] = tid + k * GT < J ? to_f32(en[tid + k * GT]) : 0.f;
    }
    // ---- joint: h = round(tanh(e_t + g)); the own rows of the output product; arg-max candidates -------------------------------
    for (int i = tid; i < J; i += GT) hj[i] = rnd<T>(tanh_t<T>(etL[i] + gL[i]));
    __syncthreads();
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int r0 = 0; r0 < nv; r0 += 32) {
      const int r = r0 + grp;
      if (r < nv) {
        const float v = rnd<T>(row_dot<T>(wo_s + (long)r * ldo, J, hj, sub) + bo[r]);
        if (v > best || (v == best && v0 + r < bi)) { best = v; bi = v0 + r; }
      }
    }
    // the 32 row groups' candidates -> one per workgroup (ties: the lowest index, as the chain's arg-max kernel)
    if (sub == 0) { red[grp] = best; reinterpret_cast<int*>(red)[32 + grp] = bi; }
    __syncthreads();
    if (tid < 64) {
      float bv = lane < 32 ? red[lane] : -INFINITY; int bx = lane < 32 ? reinterpret_cast<int*>(red)[32 + lane] : 0x7fffffff;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bx, o, 64);
        if (ov > bv || (ov == bv && oi < bx)) { bv = ov; bx = oi; }
      }
      if (lane == 0)   // (a workgroup without a finite candidate publishes index 0xFFFFF: it never wins)
        put64(a.part + (long)(step & 1) * G + g, __float_as_uint(bv), (((unsigned)(step + 1) & 0xFFFu) << 20) | ((unsigned)bx & 0xFFFFFu));
    }
    // every workgroup folds all candidates (the same data, the same order: the same decision): one wave, one candidate per lane
    if (tid < 64) {
      const unsigned long long* p = a.part + (long)(step & 1) * G;
      float bv = -INFINITY; int bx = 0x7fffffff;
      if (lane < G) {
        const unsigned long long w = get64(p + lane, 0xFFF00000ull, (unsigned long long)(((unsigned)(step + 1) & 0xFFFu) << 20), a.err);
        bv = __uint_as_float((unsigned)(w >> 32));
        bx = (int)((unsigned)w & 0xFFFFFu);
        if (bx == 0xFFFFF) { bv = -INFINITY; bx = 0x7fffffff; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bx, o, 64);
        if (ov > bv || (ov == bv && oi < bx)) { bv = ov; bx = oi; }
      }
      if (lane == 0) reinterpret_cast<int*>(red)[96] = bx == 0x7fffffff ? 0 : bx;
    }
    __syncthreads();
    const int tok = reinterpret_cast<int*>(red)[96];
    ++step;
    if (g == 0 && tid == 0) a.align[na] = tok;
    ++na;
    if (tok == a.blank) {
      ++t;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (tid + k * GT < J) etL[tid + k * GT] = enext[k];
      __syncthreads();
    } else {
      if (g == 0 && tid == 0) a.hyp[nh] = tok;
      ++nh;
      if (nh > a.max_len) break;   // (the reference stops AFTER appending the label that exceeds the limit)
      decoder_step(tok);
    }
  }
  if (g == 0 && tid == 0) { a.lens[0] = nh; a.lens[1] = na; }
}

struct GreedyState { char* buf; size_t bytes; };
GreedyState g_greedy{nullptr, 0};
int g_rnnt_greedy_coop = 1;

int greedy_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  }
  return n;
}

}  // namespace

void emo_rnnt_set_greedy_coop(int v) { g_rnnt_greedy_coop = v; }

// Can the cooperative search take this model?  (two LSTM layers, every width a multiple of 8, 32 workgroups that own whole units)
// workgroups of the search: 64 (half the rows per workgroup and phase) when every matrix has that many rows / units, else 32
static int greedy_groups(int H, int V) { return (H % 64 == 0 && V >= 64 && 4 * (H / 64) <= 64) ? 64 : 32; }

extern "C" long emoasr_rnnt_greedy_supported(int dtype, int E, int H, int J, int V, int nl) {
  const int G = greedy_groups(H, V);
  return g_rnnt_greedy_coop && (dtype == EMO_BF16 || dtype == EMO_F32) && nl == 2 && E % 8 == 0 && H % 8 == 0 && J % 8 == 0 &&
         H % G == 0 && 4 * (H / G) <= 64 && E <= G_MAXE && H <= G_MAXH && J <= 4 * GT && V >= G && J >= G;
}

// LDS bytes of one workgroup of the search (the own units' LSTM rows of both layers included when everything fits: *wlds)
static size_t greedy_smem(int dtype, int E, int H, int J, int V, bool* wlds) {
  const int G = greedy_groups(H, V);
  const size_t esz = dtype == EMO_BF16 ? 2 : 4;
  const int nvmax = (V + G - 1) / G + 1, njmax = (J + G - 1) / G + 1, nu = H / G;
  const size_t pad = 16 / esz;
  const size_t nfl = (size_t)E + 2 * H + 3 * J + 128 + 8 * nu + nvmax + njmax + 2 * nu;
  const size_t smem0 = (nfl * 4 + 15) / 16 * 16 + ((size_t)nvmax * (J + pad) + (size_t)njmax * (H + pad)) * esz;
  const size_t lstm = ((size_t)4 * nu * (E + pad + H + pad) + (size_t)4 * nu * (H + pad + H + pad)) * esz;
  *wlds = smem0 + lstm <= 160 * 1024;
  return smem0 + (*wlds ? lstm : 0);
}

// Can ONE utterance of T encoder frames with at most max_len labels go through the one-launch search?  (_supported's model
// conditions + the workgroup's LDS image within 160 KB + T + max_len within the 12-bit step tag of the hand-off words.)  Callers
// fall back to the launch chain (emoasr_joint_tanh / emoasr_argmax_rows / emoasr_first_not_equal) for an utterance that does not fit.
extern "C" long emoasr_rnnt_greedy_fits(int dtype, int E, int H, int J, int V, int nl, int T, int max_len) {
  if (!emoasr_rnnt_greedy_supported(dtype, E, H, J, V, nl)) return 0;
  bool wlds = false;
  return greedy_smem(dtype, E, H, J, V, &wlds) <= 160 * 1024 && T + max_len + 2 < 4095 && V < (1 << 20);
}

// scratch of emoasr_rnnt_greedy (bytes; no initialisation needed)
extern "C" long emoasr_rnnt_greedy_ws_bytes(int H, int J) { return (long)(2 * H + J + 2 * 64) * 8 + 256; }

// One utterance.  e_all [T, J] = w_enc . eouts + bias in the compute dtype; hyp int32 [max_len + 1], align int32 [T + max_len + 1],
// lens int32 [2] = {len(hyp), len(align)} (device).  b_lstm0 / b_lstm1: bias_ih + bias_hh of the two layers.
extern "C" int emoasr_rnnt_greedy(int dtype, int T, int E, int H, int J, int V, int blank, int eos, int max_len, const void* e_all,
                                  const void* emb, const void* w_ih0, const void* w_hh0, const float* b_lstm0, const void* w_ih1,
                                  const void* w_hh1, const float* b_lstm1, const void* w_dec, const float* b_dec, const void* w_out,
                                  const float* b_out, void* ws, long ws_bytes, int* hyp, int* align, int* lens, void* stream) {
  EMO_CHECK(emoasr_rnnt_greedy_supported(dtype, E, H, J, V, 2), "rnnt_greedy: unsupported model (dtype %d E %d H %d J %d V %d)", dtype,
            E, H, J, V);
  EMO_CHECK(ws && ws_bytes >= emoasr_rnnt_greedy_ws_bytes(H, J), "rnnt_greedy: scratch too small");
  hipStream_t s = (hipStream_t)stream;
  const int G = greedy_groups(H, V);
  if (T <= 0) { return hipMemsetAsync(lens, 0, 8, s) == hipSuccess ? 0 : 1; }
  EMO_CHECK(T + max_len + 2 < 4095 && V < (1 << 20), "rnnt_greedy: T + max_len = %d steps / V = %d exceed the 12-bit step tag / 20-bit index",
            T + max_len, V);
  const size_t esz = dtype == EMO_BF16 ? 2 : 4;
  GreedyArgs a{};
  a.T = T; a.E = E; a.H = H; a.J = J; a.V = V; a.blank = blank; a.eos = eos; a.max_len = max_len;
  a.e_all = e_all; a.emb = emb;
  a.w_ih[0] = w_ih0; a.w_hh[0] = w_hh0; a.b_lstm[0] = b_lstm0;
  a.w_ih[1] = w_ih1; a.w_hh[1] = w_hh1; a.b_lstm[1] = b_lstm1;
  a.w_dec = w_dec; a.b_dec = b_dec; a.w_out = w_out; a.b_out = b_out;
  char* p = static_cast<char*>(ws);
  a.counter = reinterpret_cast<unsigned*>(p); a.err = reinterpret_cast<int*>(p + 64);
  unsigned long long* f = reinterpret_cast<unsigned long long*>(p + 256);
  a.hbuf = f; f += 2 * H;
  a.gbuf = f; f += J;
  a.part = f;
  a.hyp = hyp; a.align = align; a.lens = lens;
  // tags start at 1 in every launch: the exchange words (and the error flag) are cleared on the launch stream first
  EMO_CHECK(hipMemsetAsync(p, 0, (size_t)emoasr_rnnt_greedy_ws_bytes(H, J), s) == hipSuccess, "rnnt_greedy: hipMemsetAsync failed");
  bool wlds = false;
  const size_t smem = greedy_smem(dtype, E, H, J, V, &wlds);
  EMO_CHECK(smem <= 160 * 1024, "rnnt_greedy: %zu bytes of LDS needed", smem);
  const void* kerns[4] = {(const void*)rnnt_greedy_kernel<bf16, false>, (const void*)rnnt_greedy_kernel<bf16, true>,
                          (const void*)rnnt_greedy_kernel<float, false>, (const void*)rnnt_greedy_kernel<float, true>};
  const int ki = (dtype == EMO_BF16 ? 0 : 2) + (wlds ? 1 : 0);
  static size_t set_bytes[4] = {0, 0, 0, 0};
  if (smem > set_bytes[ki]) {
    hipError_t e = hipFuncSetAttribute(kerns[ki], hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    EMO_CHECK(e == hipSuccess, "rnnt_greedy: hipFuncSetAttribute(%zu): %s", smem, hipGetErrorString(e));
    int per_cu = 0;   // the G workgroups must be co-resident: the launch is a plain <<<>>>
    EMO_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kerns[ki], GT, smem) == hipSuccess &&
                  (long)per_cu * greedy_cus() >= G, "rnnt_greedy: the device cannot hold %d workgroups at once", G);
    set_bytes[ki] = smem;
  }
  switch (ki) {
    case 0: rnnt_greedy_kernel<bf16, false><<<G, GT, smem, s>>>(a); break;
    case 1: rnnt_greedy_kernel<bf16, true><<<G, GT, smem, s>>>(a); break;
    case 2: rnnt_greedy_kernel<float, false><<<G, GT, smem, s>>>(a); break;
    default: rnnt_greedy_kernel<float, true><<<G, GT, smem, s>>>(a); break;
  }
  EMO_LAUNCH_CHECK();
  return 0;
}

// error flag of the last launch's barriers in `ws` (1: a wait gave up; the outputs are then undefined).  Synchronises the stream.
extern "C" long emoasr_rnnt_greedy_status(const void* ws, void* stream) {
  int e = 0;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
  if (hipMemcpy(&e, static_cast<const char*>(ws) + 64, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return e;
}
