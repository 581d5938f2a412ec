// Small-M building blocks of the cached decode steps (csrc/decode_rt.hip): at beam 10 every linear layer of the Transformer
// decoder and of the Transformer LM processes 10 rows per output step, and a step is a chain of ~100 dependent launches of
// 3-6 us each (HIP-graph replayed), so its time is the NUMBER of kernels on the longer chain.  These two kernels fold the
// LayerNorms, the cache append and the residual adds into their neighbours (bf16 only; the f32 parity mode keeps the
// separate launches):
//
//   rowlin     y[M<=16, N] = act( LNa?(x)[M, K] . W[N, K]^T + b ) + ( r | LNr(r) )        one 16x16x32 MFMA column strip per wave
//   attn_step  one query position per (hypothesis, head) against the K / V cache, appending the new key / value first
//
// Measured (MI355X, beam 10, d 256, HIP-graph timed): rowlin 6.0 us plain / 8.0 us with a LayerNorm inside / 8.7 us with a
// LayerNorm'ed residual, against 4.7 us for the 64x64-tile GEMM and 4.5 us for a LayerNorm launch; the whole step 0.74 ms
// against 0.68 ms with the separate launches -- every launch on the chain costs 4.5-6 us whatever it does, and these two
// kernels add a second dependent load phase.  Kept behind emoasr_set_option("decode_fused", 1); default off.
//
// LNa (prologue): the A rows are normalised on the way into LDS (row statistics recomputed by every workgroup: 16 rows of
// <= 1024 values) and rounded to bf16 like the stored output of a separate LayerNorm launch.  LNr (epilogue): the residual
// is LayerNorm(r) of a post-LN block whose normalised input was never stored -- again recomputed from r's rows.
#include <math.h>
#include "common.h"
#include "../../include/emoasr_hip.h"

namespace {

struct RowLinArgs {
  int M, N, K;
  const bf16* x; long ldx;              // [M, K]
  const float *lna_g, *lna_b; float lna_eps;   // prologue LayerNorm over K (NULL: none)
  const bf16* w; const float* bias;     // [N, K], [N]
  int act;
  const bf16* res; long ldres;          // residual rows [M, N] (NULL: none)
  const float *lnr_g, *lnr_b; float lnr_eps;   // the residual is LayerNorm(res) over its N columns (NULL: plain)
  bf16* y; float* y32; long ldy;        // output (one of the two)
};

constexpr int RL_MAXK = 1024;

__global__ __launch_bounds__(256) void rowlin_kernel(const RowLinArgs a) {
  __shared__ __attribute__((aligned(16))) bf16 xs[16 * (RL_MAXK + 8)];
  __shared__ float r_mean[16], r_rstd[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ldxs = a.K + 8;
  // ---- A rows -> LDS (bf16), optionally LayerNorm'ed; rows >= M are zero.  A wave owns rows wave, wave + 4, ...; all
  //      their loads (and those of the residual rows whose statistics are needed) are issued before anything is reduced:
  //      one memory round trip, not one per row and pass ----
  {
    // lane l holds elements 4l .. 4l+3 of every 256-element chunk of a row: one 8-byte load per row and chunk (guards are
    // out-of-range buffer offsets, not branches: a branch per load makes the compiler wait for each one)
    constexpr int CH = RL_MAXK / 256;
    f32x4 v[4][CH], rv[4][CH], ga[CH], gb[CH];
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(a.x), rsr = make_rsrc(a.res ? (const void*)a.res : (const void*)a.x);
    const __amdgpu_buffer_rsrc_t rsg = make_rsrc(a.lna_g ? a.lna_g : a.bias), rsb = make_rsrc(a.lna_b ? a.lna_b : a.bias);
    auto ld4 = [&](const __amdgpu_buffer_rsrc_t& rs, bool ok, long elem) {
      typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
      const u32x2_t u = __builtin_amdgcn_raw_buffer_load_b64(rs, ok ? (unsigned)(elem * 2) : EMO_OOB, 0, 0);
      return f32x4{__uint_as_float(u[0] << 16), __uint_as_float(u[0] & 0xffff0000u), __uint_as_float(u[1] << 16),
                   __uint_as_float(u[1] & 0xffff0000u)};
    };
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = wave + 4 * rr;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int k = 256 * c + 4 * lane;
        v[rr][c] = ld4(rsx, r < a.M && k < a.K, (long)r * a.ldx + k);
        rv[rr][c] = ld4(rsr, a.lnr_g && r < a.M && k < a.N, (long)r * a.ldres + k);
      }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int k = 256 * c + 4 * lane;
      const bool ok = a.lna_g && k < a.K;
      ga[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsg, ok ? (unsigned)(k * 4) : EMO_OOB, 0, 0));
      gb[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsb, ok ? (unsigned)(k * 4) : EMO_OOB, 0, 0));
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = wave + 4 * rr;
      if (a.lna_g && r < a.M) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) s += (v[rr][c][0] + v[rr][c][1]) + (v[rr][c][2] + v[rr][c][3]);
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / a.K;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          if (256 * c + 4 * lane < a.K) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[rr][c][e] - mean; q += d * d; }
          }
        }
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = rsqrtf(q / a.K + a.lna_eps);
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[rr][c][e] = (v[rr][c][e] - mean) * rstd * ga[c][e] + gb[c][e];
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int k = 256 * c + 4 * lane;
        if (k < a.K) {
          bf16x4 h;
#pragma unroll
          for (int e = 0; e < 4; ++e) h[e] = (bf16)v[rr][c][e];
          *reinterpret_cast<bf16x4*>(&xs[r * ldxs + k]) = h;
        }
      }
      if (a.lnr_g && r < a.M) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) s += (rv[rr][c][0] + rv[rr][c][1]) + (rv[rr][c][2] + rv[rr][c][3]);
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / a.N;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          if (256 * c + 4 * lane < a.N) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = rv[rr][c][e] - mean; q += d * d; }
          }
        }
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        if (lane == 0) { r_mean[r] = mean; r_rstd[r] = rsqrtf(q / a.N + a.lnr_eps); }
      }
    }
  }
  __syncthreads();
  // ---- one 16-column strip per wave: D[m][n] = sum_k x[m][k] * W[n][k] ----
  const int n0 = (blockIdx.x * 4 + wave) * 16;
  if (n0 >= a.N) return;
  const int col = n0 + (lane & 15);
  const bool cok = col < a.N;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  const __amdgpu_buffer_rsrc_t rsw = make_rsrc(a.w);
  const unsigned woff = (unsigned)(((long)(cok ? col : 0) * a.K + 8 * (lane >> 4)) * 2);
  const bf16* xrow = xs + (lane & 15) * ldxs + 8 * (lane >> 4);
  // weight fragments in groups of 8 loads issued together (a dependent load per MFMA would cost a memory round trip each)
  for (int kc = 0; kc < a.K; kc += 256) {
    bf16x8 wf[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k0 = kc + 32 * i;
      wf[i] = buf_load16<bf16>(rsw, (cok && k0 < a.K) ? woff + (unsigned)(k0 * 2) : EMO_OOB).v;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k0 = kc + 32 * i;
      if (k0 < a.K) {   // (wave-uniform)
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xrow + k0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, wf[i], acc, 0, 0, 0);
      }
    }
  }
  // acc[r] = D[row 4 * (lane >> 4) + r][col lane & 15]
  if (!cok) return;
  const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = 4 * (lane >> 4) + r;
    if (m >= a.M) continue;
    float v = apply_act(a.act, acc[r] + bv);
    if (a.res) {
      float rv = (float)a.res[(long)m * a.ldres + col];
      if (a.lnr_g) rv = (float)(bf16)((rv - r_mean[m]) * r_rstd[m] * a.lnr_g[col] + a.lnr_b[col]);
      v += rv;
    }
    if (a.y32) a.y32[(long)m * a.ldy + col] = v;
    else a.y[(long)m * a.ldy + col] = (bf16)v;
  }
}

// One query per (hypothesis b, head h): q / new k / new v are row b of qkv [nb, 3d]; the new key and value are appended
// to the caches [nb][Lmax][d] at position pos (device scalar), then softmax(q . K^T * scale) . V over positions 0 .. pos.
__global__ __launch_bounds__(64) void attn_step_kernel(int d, int dk, int Lmax, const bf16* __restrict__ qkv,
                                                       bf16* __restrict__ kc, bf16* __restrict__ vc,
                                                       const int* __restrict__ pos_p, float scale, bf16* __restrict__ out) {
  extern __shared__ float sc[];  // [Lmax] scores / probabilities
  __shared__ float qs[128];
  const int b = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
  const int pos = *pos_p;
  const bf16* row = qkv + (long)b * 3 * d + h * dk;
  bf16* kb = kc + (long)b * Lmax * d + h * dk;
  bf16* vb = vc + (long)b * Lmax * d + h * dk;
  for (int c = lane; c < dk; c += 64) {
    qs[c] = (float)row[c];
    kb[(long)pos * d + c] = row[d + c];
    vb[(long)pos * d + c] = row[2 * d + c];
  }
  __syncthreads();   // (one wave: orders the cache writes before the reads below as well)
  float mx = -INFINITY;
  for (int t = lane; t <= pos; t += 64) {
    const bf16* kr = kb + (long)t * d;
    float s = 0.f;
    for (int c = 0; c < dk; c += 8) {
      const bf16x8 kv = *reinterpret_cast<const bf16x8*>(kr + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += qs[c + e] * (float)kv[e];
    }
    s *= scale;
    sc[t] = s;
    mx = fmaxf(mx, s);
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
  for (int t = lane; t <= pos; t += 64) {
    const float p = __expf(sc[t] - mx);
    sc[t] = p;
    sum += p;
  }
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  __syncthreads();
  const float inv = 1.f / sum;
  for (int c = lane; c < dk; c += 64) {
    float o = 0.f;
    for (int t = 0; t <= pos; ++t) o += sc[t] * (float)vb[(long)t * d + c];
    out[(long)b * d + h * dk + c] = (bf16)(o * inv);
  }
}

}  // namespace

// y = act(LN?(x) . W^T + bias) (+ residual | LayerNorm(residual)); bf16, M <= 16, K <= 1024 and a multiple of 32
// (rows 8-byte aligned).
extern "C" int emoasr_rowlin(int M, int N, int K, const void* x, long ldx, const float* lna_g, const float* lna_b, float lna_eps,
                             const void* w, const float* bias, int act, const void* res, long ldres, const float* lnr_g,
                             const float* lnr_b, float lnr_eps, void* y, int out_f32, long ldy, void* stream) {
  EMO_CHECK(M >= 1 && M <= 16 && N >= 1 && K >= 32 && K <= RL_MAXK && K % 32 == 0 && ldx % 4 == 0, "rowlin: M=%d N=%d K=%d unsupported", M, N, K);
  EMO_CHECK(!lnr_g || (res && N <= RL_MAXK && N % 4 == 0 && ldres % 4 == 0), "rowlin: LayerNorm of the residual needs res and N <= %d", RL_MAXK);
  RowLinArgs a{};
  a.M = M; a.N = N; a.K = K; a.x = (const bf16*)x; a.ldx = ldx; a.lna_g = lna_g; a.lna_b = lna_b; a.lna_eps = lna_eps;
  a.w = (const bf16*)w; a.bias = bias; a.act = act; a.res = (const bf16*)res; a.ldres = ldres;
  a.lnr_g = lnr_g; a.lnr_b = lnr_b; a.lnr_eps = lnr_eps;
  if (out_f32) a.y32 = (float*)y; else a.y = (bf16*)y;
  a.ldy = ldy;
  rowlin_kernel<<<cdiv(N, 64), 256, 0, (hipStream_t)stream>>>(a);
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_attn_step(int nb, int d, int H, int Lmax, const void* qkv, void* kcache, void* vcache, const int* pos,
                                void* out, void* stream) {
  EMO_CHECK(nb >= 1 && d % H == 0 && (d / H) % 8 == 0 && d / H <= 128, "attn_step: nb=%d d=%d H=%d unsupported", nb, d, H);
  dim3 grid(nb, H);
  attn_step_kernel<<<grid, 64, Lmax * sizeof(float), (hipStream_t)stream>>>(d, d / H, Lmax, (const bf16*)qkv, (bf16*)kcache,
                                                                           (bf16*)vcache, pos, 1.f / sqrtf((float)(d / H)),
                                                                           (bf16*)out);
  EMO_LAUNCH_CHECK();
  return 0;
}
