"""On-GPU log-mel front end: host-side constants of the Kaldi filterbank recipe the reference extracts its
features with (corpora/utils/wav_to_feats.py:21-35: torchaudio.compliance.kaldi.fbank(wav * 2**15,
window_type="hamming", htk_compat=True, sample_frequency=16000, num_mel_bins=80, use_energy=False), every
other option at its default: 25 ms / 10 ms frames, snip_edges, dither 0, DC-offset removal, pre-emphasis
0.97, FFT padded to 512, mel range 20 Hz .. Nyquist, power spectrum, log with floor FLT_EPSILON) and the
wrapper that runs the HIP kernel (csrc/feats.hip: one block per frame, LDS radix-2 FFT).
"""
import math

import numpy as np
import torch

from . import ops


def hamming_window(n):
    """Kaldi "hamming": 0.54 - 0.46 cos(2 pi i / (n - 1))"""
    i = np.arange(n, dtype=np.float64)
    return 0.54 - 0.46 * np.cos(2.0 * math.pi * i / (n - 1))


def mel_scale(f):
    return 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def kaldi_mel_banks(n_mel=80, n_fft=512, sample_rate=16000.0, low_freq=20.0, high_freq=0.0):
    """[n_mel, n_fft/2 + 1] triangular filters, equally spaced on the mel scale between low_freq and
    Nyquist + high_freq; evaluated at the n_fft/2 FFT bins below Nyquist (the Nyquist column is zero),
    as Kaldi's MelBanks / torchaudio's get_mel_banks do."""
    nyq = 0.5 * sample_rate
    if high_freq <= 0.0:
        high_freq += nyq
    nb = n_fft // 2
    mel_lo, mel_hi = mel_scale(low_freq), mel_scale(high_freq)
    delta = (mel_hi - mel_lo) / (n_mel + 1)
    left = mel_lo + np.arange(n_mel)[:, None] * delta
    center, right = left + delta, left + 2.0 * delta
    mel = mel_scale(np.arange(nb) * (sample_rate / n_fft))[None, :]
    up, down = (mel - left) / (center - left), (right - mel) / (right - center)
    fb = np.maximum(0.0, np.minimum(up, down))
    return np.concatenate([fb, np.zeros((n_mel, 1))], axis=1)


class LogMel:
    """wav (float in [-1, 1), 16 kHz, on the device) -> log-mel [T, n_mel] f32 on the device"""

    def __init__(self, device, n_mel=80, sample_rate=16000, frame_ms=25.0, shift_ms=10.0, preemph=0.97):
        self.frame_len = int(sample_rate * frame_ms * 0.001)
        self.frame_shift = int(sample_rate * shift_ms * 0.001)
        self.n_fft = 1 << (self.frame_len - 1).bit_length()
        self.n_mel, self.preemph = n_mel, preemph
        self.window = torch.from_numpy(hamming_window(self.frame_len)).float().to(device)
        self.mel_fb = torch.from_numpy(kaldi_mel_banks(n_mel, self.n_fft, float(sample_rate))).float().to(device).contiguous()

    def num_frames(self, n_samples):
        return 1 + (n_samples - self.frame_len) // self.frame_shift if n_samples >= self.frame_len else 0

    def __call__(self, wav, scale=2.0 ** 15):
        wav = (wav.reshape(-1).float() * scale).contiguous()  # "wav *= 2 ** 15  # kaldi" (wav_to_feats.py:25)
        return ops.fbank(wav, self.frame_len, self.frame_shift, self.n_fft, self.n_mel, self.preemph, self.window,
                         self.mel_fb)
