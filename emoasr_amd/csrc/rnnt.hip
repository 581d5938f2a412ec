// RNN-Transducer decoder kernels: LSTM cell (prediction network), joint-network broadcast/tanh and
// its reductions, transducer lattice (alpha/beta over anti-diagonals) with the gradient w.r.t. the
// joint logits, row arg-max for greedy decoding.
//
// Reference: asr/modeling/decoders/rnn_transducer.py:81-240 (recurrency :158-192, joint :147-156,
// loss call :102-115 -> third-party warp_rnnt.rnnt_loss, greedy :194-240).  The loss follows the
// published transducer forward-backward (Graves 2012) with that call's semantics: log-softmax
// over V per (t,u) cell, blank index, mean over the batch, no frame averaging; it is checked against
// oracle/rnnt.py (path enumeration) -- warp_rnnt itself is absent, parity unpinned.
#include "common.h"
#include "../../include/emoasr_hip.h"

namespace {

// gates_pre [B,4H] (i|f|g|o pre-activations), c_prev f32 [B,H] (NULL = zeros)
template <typename T>
__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(int B, int H, const T* __restrict__ gp,
                                                            const float* __restrict__ c_prev,
                                                            T* __restrict__ h, long ldh, float* __restrict__ c,
                                                            T* __restrict__ ga) {
  const int n = B * H;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int b = i / H, j = i % H;
    const T* g4 = gp + (long)b * 4 * H;
    const float ig = sigmoid_t<T>(to_f32(g4[j])), fg = sigmoid_t<T>(to_f32(g4[H + j]));
    const float gg = tanh_t<T>(to_f32(g4[2 * H + j])), og = sigmoid_t<T>(to_f32(g4[3 * H + j]));
    const float cn = fg * (c_prev ? c_prev[i] : 0.f) + ig * gg;
    c[i] = cn;
    h[(long)b * ldh + j] = from_f32<T>(og * tanh_t<T>(cn));
    T* a4 = ga + (long)b * 4 * H;
    a4[j] = from_f32<T>(ig); a4[H + j] = from_f32<T>(fg); a4[2 * H + j] = from_f32<T>(gg); a4[3 * H + j] = from_f32<T>(og);
  }
}

// dh_out (from the layer above, row stride lddh) + dh_rec (recurrent, may be NULL) ; dc in/out f32
template <typename T>
__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(int B, int H, const T* __restrict__ dh_out, long lddh,
                                                            const T* __restrict__ dh_rec,
                                                            float* __restrict__ dc, const T* __restrict__ ga,
                                                            const float* __restrict__ c_prev,
                                                            const float* __restrict__ c, T* __restrict__ dgp) {
  const int n = B * H;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int b = i / H, j = i % H;
    const T* a4 = ga + (long)b * 4 * H;
    const float ig = to_f32(a4[j]), fg = to_f32(a4[H + j]), gg = to_f32(a4[2 * H + j]), og = to_f32(a4[3 * H + j]);
    float dh = to_f32(dh_out[(long)b * lddh + j]);
    if (dh_rec) dh += to_f32(dh_rec[i]);
    const float tc = tanh_t<T>(c[i]);
    const float dct = dc[i] + dh * og * (1.f - tc * tc);
    const float cp = c_prev ? c_prev[i] : 0.f;
    T* d4 = dgp + (long)b * 4 * H;
    d4[j] = from_f32<T>(dct * gg * ig * (1.f - ig));
    d4[H + j] = from_f32<T>(dct * cp * fg * (1.f - fg));
    d4[2 * H + j] = from_f32<T>(dct * ig * (1.f - gg * gg));
    d4[3 * H + j] = from_f32<T>(dh * tc * og * (1.f - og));
    dc[i] = dct * fg;
  }
}

// h[b,t,u,:] = tanh(e[b,t,:] + g[b,u,:])
template <typename T>
__global__ __launch_bounds__(256) void joint_tanh_kernel(long n, int Tn, int U, int J, const T* __restrict__ e,
                                                         const T* __restrict__ g, T* __restrict__ h) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int j = i % J;
    long r = i / J;
    const int u = r % U; r /= U;
    const int t = r % Tn;
    const long b = r / Tn;
    h[i] = from_f32<T>(tanhf(to_f32(e[(b * Tn + t) * J + j]) + to_f32(g[(b * U + u) * J + j])));
  }
}
// bf16, J % 8 == 0: eight values per thread (16-byte loads / stores; the element-wise kernel wrote 2 bytes per thread and spent
// most of its 0.5 ms in tanhf) with tanh(x) = 1 - 2 / (exp(2x) + 1) on the hardware exp / rcp units (relative error ~1e-6, below
// the output's bf16 rounding)
__global__ __launch_bounds__(256) void joint_tanh8_kernel(long n8, int Tn, int U, int J8, const bf16* __restrict__ e,
                                                          const bf16* __restrict__ g, bf16* __restrict__ h) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const int j = i % J8;
    long r = i / J8;
    const int u = r % U; r /= U;
    const int t = r % Tn;
    const long b = r / Tn;
    const bf16x8 ev = *reinterpret_cast<const bf16x8*>(e + ((b * Tn + t) * J8 + j) * 8);
    const bf16x8 gv = *reinterpret_cast<const bf16x8*>(g + ((b * U + u) * J8 + j) * 8);
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float x = (float)ev[k] + (float)gv[k];
      const float ex = __expf(2.f * x);                       // inf for large x -> 1 - 0 = 1; 0 for very negative x -> 1 - 2 = -1
      o[k] = (bf16)(1.f - 2.f * __builtin_amdgcn_rcpf(ex + 1.f));
    }
    *reinterpret_cast<bf16x8*>(h + i * 8) = o;
  }
}
// mode 0: de[b,t,j] = sum_u d[b,t,u,j];  mode 1: dg[b,u,j] = sum_t d[b,t,u,j]
// J % 8 == 0: eight columns per thread (16-byte loads), four terms of the sum in flight; the terms are added in the same order as
// the element-wise kernel below adds them, so the two agree bit for bit.  (One 2-byte load per thread and term ran at 2.3 TB/s.)
__global__ __launch_bounds__(256) void joint_reduce8_kernel(int mode, int Bn, int Tn, int U, int J, const bf16* __restrict__ d,
                                                            bf16* __restrict__ out) {
  const int j8 = J / 8;
  const long n = (long)Bn * (mode == 0 ? Tn : U) * j8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int j = (int)(i % j8) * 8;
    const long r = i / j8;
    const bf16* p;
    long step;
    int cnt;
    if (mode == 0) {
      const int t = (int)(r % Tn); const long b = r / Tn;
      p = d + ((b * Tn + t) * U) * J + j; step = J; cnt = U;
    } else {
      const int u = (int)(r % U); const long b = r / U;
      p = d + ((b * Tn) * U + u) * J + j; step = (long)U * J; cnt = Tn;
    }
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    int k = 0;
    for (; k + 4 <= cnt; k += 4) {
      float x[4][8];
#pragma unroll
      for (int q = 0; q < 4; ++q) load8<bf16>(p + (long)(k + q) * step, x[q]);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += x[q][e];
    }
    for (; k < cnt; ++k) {
      float x[8];
      load8<bf16>(p + (long)k * step, x);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += x[e];
    }
    store8<bf16>(out + i * 8, s);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void joint_reduce_kernel(int mode, int Bn, int Tn, int U, int J,
                                                           const T* __restrict__ d, T* __restrict__ out) {
  const long n = (long)Bn * (mode == 0 ? Tn : U) * J;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int j = i % J;
    long r = i / J;
    float s = 0.f;
    if (mode == 0) {
      const int t = r % Tn; const long b = r / Tn;
      const T* p = d + ((b * Tn + t) * U) * J + j;
      for (int u = 0; u < U; ++u) s += to_f32(p[(long)u * J]);
    } else {
      const int u = r % U; const long b = r / U;
      const T* p = d + ((b * Tn) * U + u) * J + j;
      for (int t = 0; t < Tn; ++t) s += to_f32(p[(long)t * U * J]);
    }
    out[i] = from_f32<T>(s);
  }
}

// one wave per (b,t,u) row: lse over V, blank and label log-probs
template <typename T>
__global__ __launch_bounds__(256) void rnnt_gather_kernel(long rows, int Tn, int U, int V, int Lmax,
                                                          const T* __restrict__ z, const int* __restrict__ labels,
                                                          const int* __restrict__ ylens, int blank,
                                                          float* __restrict__ lse, float* __restrict__ lpb,
                                                          float* __restrict__ lpy) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* zr = z + row * V;
  float m = -INFINITY, s = 0.f;
  constexpr int VEC = 16 / sizeof(T), NV = 4;   // a row of up to 64 * NV * VEC values in registers: read once, 16 bytes per load
  if (V % VEC == 0 && V <= 64 * NV * VEC) {
    // (two passes of 2-byte loads in run-time loops -- one dependent round trip per 64 values -- ran at 1.5 TB/s)
    float x[NV][VEC];
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(zr);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v0 = (lane + 64 * i) * VEC;
      const Vec16<T> q = buf_load16<T>(rs, v0 < V ? (unsigned)(v0 * sizeof(T)) : EMO_OOB);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        x[i][e] = v0 < V ? q.get(e) : -INFINITY;
        m = fmaxf(m, x[i][e]);
      }
    }
    m = wave_max(m);
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int e = 0; e < VEC; ++e) s += __expf(x[i][e] - m);   // exp(-inf) = 0 past the row's end
  } else {
    for (int v = lane; v < V; v += 64) m = fmaxf(m, to_f32(zr[v]));
    m = wave_max(m);
    for (int v = lane; v < V; v += 64) s += __expf(to_f32(zr[v]) - m);
  }
  s = wave_sum(s);
  if (lane == 0) {
    const float l = m + logf(s);
    const int u = row % U;
    const long b = row / ((long)Tn * U);
    lse[row] = l;
    lpb[row] = to_f32(zr[blank]) - l;
    lpy[row] = u < ylens[b] ? to_f32(zr[labels[b * Lmax + u]]) - l : -INFINITY;
  }
}

// log-space sum on the hardware exp2 / log2 units (same helper as the CTC lattice, ctc.hip): the recursion is
// one dependent chain per anti-diagonal and the accurate expf / log1pf expansions were most of a step
__device__ __forceinline__ float lat_add(float a, float b) {
  const float m = fmaxf(a, b);
  if (m == -INFINITY) return -INFINITY;
  return m + __logf(1.f + __expf(-fabsf(a - b)));
}

// blocks [0,B): alpha; [B,2B): beta.  Threads = label positions u; anti-diagonals d = t + u.
__global__ __launch_bounds__(1024) void rnnt_lattice_kernel(int Bn, int Tn, int U, const float* __restrict__ lpb,
                                                            const float* __restrict__ lpy,
                                                            const int* __restrict__ elens,
                                                            const int* __restrict__ ylens,
                                                            float* __restrict__ alpha, float* __restrict__ beta,
                                                            float* __restrict__ nll) {
  extern __shared__ float sh[];  // [2][U+2]
  const bool fwd = blockIdx.x < Bn;
  const int b = fwd ? blockIdx.x : blockIdx.x - Bn;
  const int u = threadIdx.x;
  const int T = min(elens[b], Tn), Ub = ylens[b];  // valid cells: t < T, u <= Ub
  const float* pb = lpb + (long)b * Tn * U;
  const float* py = lpy + (long)b * Tn * U;
  float* out = (fwd ? alpha : beta) + (long)b * Tn * U;
  float* cur = sh + 1;
  float* prv = sh + (U + 2) + 1;
  if (T <= 0) { if (fwd && u == 0) nll[b] = INFINITY; return; }
  for (int i = threadIdx.x; i < 2 * (U + 2); i += blockDim.x) sh[i] = -INFINITY;
  __syncthreads();
  float own = -INFINITY;  // this thread's value on its previous diagonal (alpha[t-1,u] / beta[t+1,u])
  const int nd = T + Ub;  // diagonals 0 .. T+Ub-1
  // The blank / label log-probabilities a cell needs do not depend on the recursion: they are fetched CH
  // diagonals ahead (two register sets), so a diagonal costs an LDS exchange + barrier instead of two
  // dependent global loads (~1.5 us per diagonal before).
  constexpr int CH = 8;
  float sA[CH], eA[CH], sB[CH], eB[CH];  // stay / emit terms of the current and the next chunk
  auto fetch = [&](int d0, float* st, float* em) {
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int d = d0 + k;
      float a = -INFINITY, e = -INFINITY;
      if (d < nd && u <= Ub) {
        if (fwd) {
          const int t = d - u;
          if (t >= 0 && t < T) {
            if (t > 0) a = pb[(long)(t - 1) * U + u];
            if (u > 0) e = py[(long)t * U + u - 1];
          }
        } else {
          const int t = T - 1 - (d - (Ub - u));
          if (t >= 0 && t < T && d - (Ub - u) >= 0) {
            if (t == T - 1 && u == Ub) a = pb[(long)t * U + u];  // the final blank (used as the start value)
            else {
              if (t < T - 1) a = pb[(long)t * U + u];
              if (u < Ub) e = py[(long)t * U + u];
            }
          }
        }
      }
      st[k] = a; em[k] = e;
    }
  };
  fetch(0, sA, eA);
  for (int d0 = 0; d0 < nd; d0 += CH) {
    fetch(d0 + CH, sB, eB);
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int d = d0 + k;
      if (d >= nd) break;
      float v = -INFINITY;
      bool act = false;
      if (fwd) {
        const int t = d - u;
        if (u <= Ub && t >= 0 && t < T) {
          act = true;
          if (t == 0 && u == 0) v = 0.f;
          else {
            const float stay = t > 0 ? own + sA[k] : -INFINITY;
            const float emit = u > 0 ? prv[u - 1] + eA[k] : -INFINITY;
            v = lat_add(stay, emit);
          }
          out[(long)t * U + u] = v;
        }
      } else {
        // reversed diagonal: t = T-1 - (d - (Ub - u))
        const int t = T - 1 - (d - (Ub - u));
        if (u <= Ub && t >= 0 && t < T && d - (Ub - u) >= 0) {
          act = true;
          if (t == T - 1 && u == Ub) v = sA[k];
          else {
            const float stay = t < T - 1 ? own + sA[k] : -INFINITY;
            const float emit = u < Ub ? prv[u + 1] + eA[k] : -INFINITY;
            v = lat_add(stay, emit);
          }
          out[(long)t * U + u] = v;
        }
      }
      if (u < U) cur[u] = act ? v : -INFINITY;
      if (act) own = v;
      lds_barrier();   // (the lattice values written to global memory are read back only after the loop)
      float* tmp = cur; cur = prv; prv = tmp;
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) { sA[k] = sB[k]; eA[k] = eB[k]; }
  }
  __syncthreads();   // thread 0 reads the last cell another thread stored
  if (fwd && u == 0) nll[b] = -(out[(long)(T - 1) * U + Ub] + pb[(long)(T - 1) * U + Ub]);
}

// one wave per row: dz = gs * (p * occ - [v=blank] gb - [v=y] gy), zero outside the valid lattice
template <typename T>
__global__ __launch_bounds__(256) void rnnt_grad_kernel(long rows, int Tn, int U, int V, int Lmax,
                                                        const T* __restrict__ z, const float* __restrict__ lse,
                                                        const float* __restrict__ lpb, const float* __restrict__ lpy,
                                                        const float* __restrict__ alpha, const float* __restrict__ beta,
                                                        const int* __restrict__ labels, const int* __restrict__ elens,
                                                        const int* __restrict__ ylens, const float* __restrict__ nll,
                                                        int blank, float gscale, const float* __restrict__ gscale_dev,
                                                        T* __restrict__ dz) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int u = row % U;
  const int t = (row / U) % Tn;
  const long b = row / ((long)Tn * U);
  const int Tb = min(elens[b], Tn), Ub = ylens[b];
  T* dr = dz + row * V;
  const float nl = nll[b];
  constexpr int VEC = 16 / sizeof(T);
  const bool vec = V % VEC == 0;   // rows are 16-byte aligned: 16 bytes per load / store instead of one value
  if (t >= Tb || u > Ub || !isfinite(nl)) {
    if (vec) {
      Vec16<T> zv;
      zv.zero();
      for (int v0 = lane * VEC; v0 < V; v0 += 64 * VEC) store16(dr + v0, zv);
    } else {
      for (int v = lane; v < V; v += 64) dr[v] = from_f32<T>(0.f);
    }
    return;
  }
  const float gs = gscale_dev ? gscale * gscale_dev[0] : gscale;
  const float a = alpha[row];
  float gb = 0.f, gy = 0.f;
  if (t == Tb - 1) { if (u == Ub) gb = __expf(a + lpb[row] + nl); }
  else gb = __expf(a + lpb[row] + beta[row + U] + nl);
  int y = -1;
  if (u < Ub) { y = labels[b * Lmax + u]; gy = __expf(a + lpy[row] + beta[row + 1] + nl); }
  const float occ = gb + gy;
  const float l = lse[row];
  const T* zr = z + row * V;
  if (vec) {
#pragma unroll 2
    for (int v0 = lane * VEC; v0 < V; v0 += 64 * VEC) {
      const Vec16<T> q = load16(zr + v0);
      Vec16<T> o;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const int v = v0 + e;
        float g = __expf(q.get(e) - l) * occ;
        if (v == blank) g -= gb;
        if (v == y) g -= gy;
        o.set(e, g * gs);
      }
      store16(dr + v0, o);
    }
    return;
  }
  for (int v = lane; v < V; v += 64) {
    float g = __expf(to_f32(zr[v]) - l) * occ;
    if (v == blank) g -= gb;
    if (v == y) g -= gy;
    dr[v] = from_f32<T>(g * gs);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void argmax_rows_kernel(int V, const T* __restrict__ x, long ldx,
                                                          int* __restrict__ out) {
  __shared__ float rv[4];
  __shared__ int ri[4];
  const T* row = x + (long)blockIdx.x * ldx;
  float m = -INFINITY; int mi = 0x7fffffff;
  for (int v = threadIdx.x; v < V; v += 256) {
    const float c = to_f32(row[v]);
    if (c > m || (c == m && v < mi)) { m = c; mi = v; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o, 64);
    const int oi = __shfl_xor(mi, o, 64);
    if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { rv[threadIdx.x >> 6] = m; ri[threadIdx.x >> 6] = mi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (rv[w] > m || (rv[w] == m && ri[w] < mi)) { m = rv[w]; mi = ri[w]; }
    out[blockIdx.x] = mi == 0x7fffffff ? 0 : mi;
  }
}

// out[0] = first i < n with x[i] != value (or -1), out[1] = x[that i] (or value): one wave, the transducer's
// windowed greedy search reads both with a single 8-byte D2H copy.
__global__ __launch_bounds__(64) void first_not_equal_kernel(int n, const int* __restrict__ x, int value,
                                                             int* __restrict__ out) {
  int best = 0x7fffffff;
  for (int i = threadIdx.x; i < n; i += 64)
    if (x[i] != value) { best = i; break; }  // per lane: its first hit (lanes stride the array)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o, 64));
  if (threadIdx.x == 0) {
    out[0] = best == 0x7fffffff ? -1 : best;
    out[1] = best == 0x7fffffff ? value : x[best];
  }
}

inline int ew_grid(long n) { long b = (n + 255) / 256; return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

}  // namespace

extern "C" int emoasr_lstm_cell_fwd(int dtype, int B, int H, const void* gates_pre, const float* c_prev, void* h,
                                    long ldh, float* c, void* gates_act, void* stream) {
  if (B == 0) return 0;
  EMO_DISPATCH(dtype, (lstm_cell_fwd_kernel<T><<<ew_grid((long)B * H), 256, 0, (hipStream_t)stream>>>(
                          B, H, (const T*)gates_pre, c_prev, (T*)h, ldh, c, (T*)gates_act)));
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_lstm_cell_bwd(int dtype, int B, int H, const void* dh_out, long lddh, const void* dh_rec,
                                    float* dc, const void* gates_act, const float* c_prev, const float* c,
                                    void* dgates_pre, void* stream) {
  if (B == 0) return 0;
  EMO_DISPATCH(dtype, (lstm_cell_bwd_kernel<T><<<ew_grid((long)B * H), 256, 0, (hipStream_t)stream>>>(
                          B, H, (const T*)dh_out, lddh, (const T*)dh_rec, dc, (const T*)gates_act, c_prev, c,
                          (T*)dgates_pre)));
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_joint_tanh(int dtype, int B, int T_, int U, int J, const void* e, const void* g, void* h,
                                 void* stream) {
  const long n = (long)B * T_ * U * J;
  if (n == 0) return 0;
  if (dtype == EMO_BF16 && J % 8 == 0) {
    joint_tanh8_kernel<<<ew_grid(n / 8), 256, 0, (hipStream_t)stream>>>(n / 8, T_, U, J / 8, (const bf16*)e, (const bf16*)g, (bf16*)h);
    EMO_LAUNCH_CHECK();
    return 0;
  }
  EMO_DISPATCH(dtype, (joint_tanh_kernel<T><<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(n, T_, U, J, (const T*)e,
                                                                                        (const T*)g, (T*)h)));
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_joint_reduce(int dtype, int B, int T_, int U, int J, const void* d, void* de, void* dg,
                                   void* stream) {
  if ((long)B * T_ * U * J == 0) return 0;
  if (dtype == EMO_BF16 && J % 8 == 0 && (((uintptr_t)d | (uintptr_t)de | (uintptr_t)dg) & 15) == 0) {
    joint_reduce8_kernel<<<ew_grid((long)B * T_ * J / 8), 256, 0, (hipStream_t)stream>>>(0, B, T_, U, J, (const bf16*)d, (bf16*)de);
    joint_reduce8_kernel<<<ew_grid((long)B * U * J / 8), 256, 0, (hipStream_t)stream>>>(1, B, T_, U, J, (const bf16*)d, (bf16*)dg);
    EMO_LAUNCH_CHECK();
    return 0;
  }
  EMO_DISPATCH(dtype, {
    joint_reduce_kernel<T><<<ew_grid((long)B * T_ * J), 256, 0, (hipStream_t)stream>>>(0, B, T_, U, J, (const T*)d, (T*)de);
    joint_reduce_kernel<T><<<ew_grid((long)B * U * J), 256, 0, (hipStream_t)stream>>>(1, B, T_, U, J, (const T*)d, (T*)dg);
  });
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_rnnt_forward(int dtype, int B, int T_, int U, int V, int Lmax, const void* logits,
                                   const int* labels, const int* elens, const int* ylens, int blank, float* lse,
                                   float* lpb, float* lpy, float* alpha, float* beta, float* nll, void* stream) {
  const long rows = (long)B * T_ * U;
  if (rows == 0) return 0;
  EMO_CHECK(U <= 1024, "rnnt: U=%d exceeds 1024 label positions", U);
  hipStream_t s = (hipStream_t)stream;
  EMO_DISPATCH(dtype, (rnnt_gather_kernel<T><<<cdiv(rows, 4), 256, 0, s>>>(rows, T_, U, V, Lmax, (const T*)logits, labels,
                                                                          ylens, blank, lse, lpb, lpy)));
  rnnt_lattice_kernel<<<2 * B, cdiv(U, 64) * 64, sizeof(float) * 2 * (U + 2), s>>>(B, T_, U, lpb, lpy, elens, ylens, alpha,
                                                                                   beta, nll);
  EMO_LAUNCH_CHECK();
  return 0;
}

// lse / lpb / lpy of every lattice cell from the output layer's per-chunk partials (emoasr_rnnt_head_fwd); zb / zy come in as the
// raw blank / label logits and leave as log-probabilities
__global__ __launch_bounds__(256) void rnnt_parts_kernel(long rows, int Tn, int U, int nchunk, const float* __restrict__ part,
                                                         const int* __restrict__ ylens, float* __restrict__ lse,
                                                         float* __restrict__ zb, float* __restrict__ zy) {
  const long row = (long)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const float2* pp = reinterpret_cast<const float2*>(part) + row;   // chunk-major table [nchunk][rows][2]
  float m = -INFINITY, s = 0.f;
  for (int c = 0; c < nchunk; ++c) {
    const float2 v = pp[(long)c * rows];
    const float mn = fmaxf(m, v.x);
    s = s * __expf(m - mn) + v.y * __expf(v.x - mn);
    m = mn;
  }
  const float l = m + logf(s);
  const int u = row % U;
  const long b = row / ((long)Tn * U);
  lse[row] = l;
  zb[row] = zb[row] - l;
  zy[row] = u < ylens[b] ? zy[row] - l : -INFINITY;
}

// label column of every lattice cell (b, t, u): labels[b, u] for u < ylens[b], else -1 (what emoasr_rnnt_head_fwd gathers)
__global__ __launch_bounds__(256) void rnnt_ycol_kernel(long rows, int Tn, int U, int Lmax, const int* __restrict__ labels,
                                                        const int* __restrict__ ylens, int* __restrict__ ycol) {
  const long row = (long)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const int u = row % U;
  const long b = row / ((long)Tn * U);
  ycol[row] = u < ylens[b] ? labels[b * Lmax + u] : -1;
}

extern "C" int emoasr_rnnt_ycol(int B, int T_, int U, int Lmax, const int* labels, const int* ylens, int* ycol, void* stream) {
  const long rows = (long)B * T_ * U;
  if (rows == 0) return 0;
  rnnt_ycol_kernel<<<cdiv(rows, 256), 256, 0, (hipStream_t)stream>>>(rows, T_, U, Lmax, labels, ylens, ycol);
  EMO_LAUNCH_CHECK();
  return 0;
}

// emoasr_rnnt_forward without the logits: part / zb / zy from emoasr_rnnt_head_fwd (zb, zy become lpb, lpy in place)
extern "C" int emoasr_rnnt_forward_parts(int B, int T_, int U, int V, const float* part, const int* elens, const int* ylens,
                                         float* lse, float* zb_lpb, float* zy_lpy, float* alpha, float* beta, float* nll,
                                         void* stream) {
  const long rows = (long)B * T_ * U;
  if (rows == 0) return 0;
  EMO_CHECK(U <= 1024, "rnnt: U=%d exceeds 1024 label positions", U);
  hipStream_t s = (hipStream_t)stream;
  rnnt_parts_kernel<<<cdiv(rows, 256), 256, 0, s>>>(rows, T_, U, cdiv(V, 64), part, ylens, lse, zb_lpb, zy_lpy);
  rnnt_lattice_kernel<<<2 * B, cdiv(U, 64) * 64, sizeof(float) * 2 * (U + 2), s>>>(B, T_, U, zb_lpb, zy_lpy, elens, ylens, alpha,
                                                                                   beta, nll);
  EMO_LAUNCH_CHECK();
  return 0;
}

// row constants of the output layer's gradient (what rnnt_grad_kernel forms per row): coef[n] = (lse, occ, gamma_blank,
// gamma_label) * (1, gs, gs, gs), ycol[n] = the label column or -1; all zero outside (elens, ylens) and for infeasible utterances
__global__ __launch_bounds__(256) void rnnt_coef_kernel(long rows, int Tn, int U, int Lmax, const float* __restrict__ lse,
                                                        const float* __restrict__ lpb, const float* __restrict__ lpy,
                                                        const float* __restrict__ alpha, const float* __restrict__ beta,
                                                        const int* __restrict__ labels, const int* __restrict__ elens,
                                                        const int* __restrict__ ylens, const float* __restrict__ nll,
                                                        float gscale, const float* __restrict__ gscale_dev,
                                                        float* __restrict__ coef, int* __restrict__ ycol) {
  const long row = (long)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const int u = row % U;
  const int t = (row / U) % Tn;
  const long b = row / ((long)Tn * U);
  const int Tb = min(elens[b], Tn), Ub = ylens[b];
  const float nl = nll[b];
  float occ = 0.f, gb = 0.f, gy = 0.f;
  int y = -1;
  if (t < Tb && u <= Ub && isfinite(nl)) {
    const float gs = gscale_dev ? gscale * gscale_dev[0] : gscale;
    const float a = alpha[row];
    if (t == Tb - 1) { if (u == Ub) gb = __expf(a + lpb[row] + nl); }
    else gb = __expf(a + lpb[row] + beta[row + U] + nl);
    if (u < Ub) { y = labels[b * Lmax + u]; gy = __expf(a + lpy[row] + beta[row + 1] + nl); }
    occ = (gb + gy) * gs; gb *= gs; gy *= gs;
  }
  *reinterpret_cast<f32x4*>(coef + row * 4) = f32x4{lse[row], occ, gb, gy};
  ycol[row] = y;
}

extern "C" int emoasr_rnnt_coef(int B, int T_, int U, int Lmax, const float* lse, const float* lpb, const float* lpy,
                                const float* alpha, const float* beta, const int* labels, const int* elens, const int* ylens,
                                const float* nll, float gscale, const float* gscale_dev, float* coef, int* ycol, void* stream) {
  const long rows = (long)B * T_ * U;
  if (rows == 0) return 0;
  rnnt_coef_kernel<<<cdiv(rows, 256), 256, 0, (hipStream_t)stream>>>(rows, T_, U, Lmax, lse, lpb, lpy, alpha, beta, labels, elens,
                                                                    ylens, nll, gscale, gscale_dev, coef, ycol);
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_rnnt_grad(int dtype, int B, int T_, int U, int V, int Lmax, const void* logits, const float* lse,
                                const float* lpb, const float* lpy, const float* alpha, const float* beta,
                                const int* labels, const int* elens, const int* ylens, const float* nll, int blank,
                                float gscale, const float* gscale_dev, void* dlogits, void* stream) {
  const long rows = (long)B * T_ * U;
  if (rows == 0) return 0;
  EMO_DISPATCH(dtype, (rnnt_grad_kernel<T><<<cdiv(rows, 4), 256, 0, (hipStream_t)stream>>>(
                          rows, T_, U, V, Lmax, (const T*)logits, lse, lpb, lpy, alpha, beta, labels, elens, ylens, nll,
                          blank, gscale, gscale_dev, (T*)dlogits)));
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_first_not_equal(int n, const int* x, int value, int* out, void* stream) {
  first_not_equal_kernel<<<1, 64, 0, (hipStream_t)stream>>>(n, x, value, out);
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_argmax_rows(int dtype, int M, int V, const void* x, long ldx, int* out, void* stream) {
  if (M == 0) return 0;
  EMO_DISPATCH(dtype, (argmax_rows_kernel<T><<<M, 256, 0, (hipStream_t)stream>>>(V, (const T*)x, ldx, out)));
  EMO_LAUNCH_CHECK();
  return 0;
}
