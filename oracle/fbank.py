"""CPU oracle, part 5: Kaldi-compatible log-mel filterbank in float64 numpy.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference calls torchaudio.compliance.kaldi.fbank (corpora/utils/wav_to_feats.py:26-33);
torchaudio is a third-party dependency that is neither vendored in /root/reference nor installed here
(version unpinned by the reference), so no golden vector from the reference itself exists.  This file restates
the published Kaldi recipe that function implements (kaldi/src/feat/feature-window.cc ProcessWindow,
mel-computations.cc MelBanks, feature-fbank.cc) with the reference's call-site options: hamming window, 25/10 ms,
snip_edges, dither 0, remove_dc_offset, pre-emphasis 0.97 with the first sample replicated, FFT 512,
80 bins from 20 Hz to Nyquist, power spectrum, log floored at FLT_EPSILON.

Cross-check (round 5; not a pin to the reference's own dependency): tests/test_fbank.py holds this file to
`transformers.audio_utils` -- an independent, published implementation of the same recipe (the torchaudio-free path of the
Hugging Face feature extractors) called with the options above -- to 1e-5 on the log-mel values (measured 1e-6).
"""
import numpy as np


def fbank(wav, sample_rate=16000, n_mel=80, frame_len=400, frame_shift=160, preemph=0.97, low_freq=20.0):
    wav = np.asarray(wav, dtype=np.float64)
    n_fft = 1 << (frame_len - 1).bit_length()
    T = 1 + (len(wav) - frame_len) // frame_shift if len(wav) >= frame_len else 0
    i = np.arange(frame_len)
    window = 0.54 - 0.46 * np.cos(2.0 * np.pi * i / (frame_len - 1))
    # mel banks
    mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)
    lo, hi = mel(low_freq), mel(0.5 * sample_rate)
    delta = (hi - lo) / (n_mel + 1)
    bins_mel = mel(np.arange(n_fft // 2) * (sample_rate / n_fft))
    fb = np.zeros((n_mel, n_fft // 2 + 1))
    for b in range(n_mel):
        left, center, right = lo + b * delta, lo + (b + 1) * delta, lo + (b + 2) * delta
        for k, m in enumerate(bins_mel):
            if left < m < right:
                fb[b, k] = (m - left) / (center - left) if m <= center else (right - m) / (right - center)
    out = np.zeros((T, n_mel))
    for t in range(T):
        fr = wav[t * frame_shift: t * frame_shift + frame_len].copy()
        fr -= fr.mean()
        fr = np.concatenate([[fr[0] - preemph * fr[0]], fr[1:] - preemph * fr[:-1]])
        spec = np.fft.rfft(fr * window, n_fft)
        out[t] = np.log(np.maximum(fb @ (spec.real ** 2 + spec.imag ** 2), np.finfo(np.float32).eps))
    return out
