// On-GPU input preprocessing: SpecAugment band masking, CMVN, Kaldi-style log-mel fbank.
// Reference: asr/spec_augment.py:39-95 (mask geometry is sampled on the host with the
// reference's rule and applied here), corpora/utils/wav_to_feats.py:26-33 +
// norm_feats.py:9-13 (torchaudio.compliance.kaldi.fbank defaults + CMVN).
#include "common.h"
#include "../../include/emoasr_hip.h"

namespace {

// spans[b, m, 2] = (start, end): m < nf are frequency bands, the rest time bands.
__global__ __launch_bounds__(256) void specaug_kernel(int Tn, int F, float* __restrict__ x,
                                                      const int* __restrict__ spans, int nf, int nt,
                                                      const int* __restrict__ xlens,
                                                      const float* __restrict__ fill) {
  const int b = blockIdx.y;
  const int len = xlens ? min(xlens[b], Tn) : Tn;
  const int* sp = spans + (long)b * (nf + nt) * 2;
  const float fv = fill ? fill[b] : 0.f;
  const long n = (long)len * F;
  float* xb = x + (long)b * Tn * F;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int t = i / F, f = i % F;
    bool hit = false;
    for (int m = 0; m < nf; ++m) hit |= (f >= sp[2 * m] && f < sp[2 * m + 1]);
    for (int m = nf; m < nf + nt; ++m) hit |= (t >= sp[2 * m] && t < sp[2 * m + 1]);
    if (hit) xb[i] = fv;
  }
}

__global__ __launch_bounds__(256) void cmvn_kernel(long n, int F, float* __restrict__ x,
                                                   const float* __restrict__ mean,
                                                   const float* __restrict__ stdv) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int f = i % F;
    x[i] = (x[i] - mean[f]) / stdv[f];
  }
}

// One block per frame: DC removal, pre-emphasis, window, zero-pad, radix-2 FFT in LDS,
// power spectrum, mel filterbank, log.
__global__ __launch_bounds__(256) void fbank_kernel(const float* __restrict__ wav, int frame_len,
                                                    int frame_shift, int n_fft, int log2_fft, int n_mel,
                                                    float preemph, const float* __restrict__ window,
                                                    const float* __restrict__ mel_fb,
                                                    float* __restrict__ feats) {
  extern __shared__ float sh[];  // re[n_fft], im[n_fft], red[16]
  float* re = sh;
  float* im = sh + n_fft;
  float* red = sh + 2 * n_fft;
  const int t = blockIdx.x;
  const float* fr = wav + (long)t * frame_shift;
  float s = 0.f;
  for (int i = threadIdx.x; i < frame_len; i += 256) s += fr[i];
  const float dc = block_sum(s, red) / frame_len;
  for (int i = threadIdx.x; i < n_fft; i += 256) {
    float v = 0.f;
    if (i < frame_len) {
      const float cur = fr[i] - dc;
      const float prev = fr[i > 0 ? i - 1 : 0] - dc;
      v = (cur - preemph * prev) * window[i];
    }
    // bit-reversed placement
    const int j = __brev((unsigned)i) >> (32 - log2_fft);
    re[j] = v; im[j] = 0.f;
  }
  __syncthreads();
  for (int sz = 2; sz <= n_fft; sz <<= 1) {
    const int half = sz >> 1;
    for (int k = threadIdx.x; k < n_fft / 2; k += 256) {
      const int grp = k / half, pos = k % half;
      const int i0 = grp * sz + pos, i1 = i0 + half;
      float sn, cs;
      sincosf(-2.f * 3.14159265358979323846f * pos / sz, &sn, &cs);
      const float tr = re[i1] * cs - im[i1] * sn, ti = re[i1] * sn + im[i1] * cs;
      const float ur = re[i0], ui = im[i0];
      re[i0] = ur + tr; im[i0] = ui + ti;
      re[i1] = ur - tr; im[i1] = ui - ti;
    }
    __syncthreads();
  }
  const int nb = n_fft / 2 + 1;
  for (int k = threadIdx.x; k < nb; k += 256) re[k] = re[k] * re[k] + im[k] * im[k];
  __syncthreads();
  for (int m = threadIdx.x; m < n_mel; m += 256) {
    float e = 0.f;
    const float* w = mel_fb + (long)m * nb;
    for (int k = 0; k < nb; ++k) e += w[k] * re[k];
    feats[(long)t * n_mel + m] = logf(fmaxf(e, 1.1920928955078125e-07f));
  }
}

inline int ew_grid(long n) { long b = (n + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

}  // namespace

extern "C" int emoasr_specaug_apply(int B, int Tn, int F, float* x, const int* spans, int nf, int nt,
                                    const int* xlens, const float* fill, void* stream) {
  if (B == 0 || Tn == 0 || nf + nt == 0) return 0;
  dim3 grid(ew_grid((long)Tn * F), B);
  specaug_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(Tn, F, x, spans, nf, nt, xlens, fill);
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_cmvn(int M, int F, float* x, const float* mean, const float* stdv, void* stream) {
  const long n = (long)M * F;
  if (n == 0) return 0;
  cmvn_kernel<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(n, F, x, mean, stdv);
  EMO_LAUNCH_CHECK();
  return 0;
}

extern "C" int emoasr_fbank(const float* wav, long n_samples, int frame_len, int frame_shift, int n_fft,
                            int n_mel, float preemph, const float* window, const float* mel_fb,
                            float* feats, int Tn, void* stream) {
  if (Tn == 0) return 0;
  int lg = 0;
  while ((1 << lg) < n_fft) ++lg;
  EMO_CHECK((1 << lg) == n_fft && n_fft >= frame_len, "fbank: n_fft=%d must be a power of two >= frame_len", n_fft);
  EMO_CHECK((long)(Tn - 1) * frame_shift + frame_len <= n_samples, "fbank: T=%d frames exceed the signal", Tn);
  fbank_kernel<<<Tn, 256, sizeof(float) * (2 * n_fft + 16), (hipStream_t)stream>>>(
      wav, frame_len, frame_shift, n_fft, lg, n_mel, preemph, window, mel_fb, feats);
  EMO_LAUNCH_CHECK();
  return 0;
}
