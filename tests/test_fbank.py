"""Log-mel front end: the host constants against the oracle's independent restatement (CPU), the HIP kernel
against the oracle (GPU).  Parity with the reference's torchaudio call is UNPINNED (see oracle/fbank.py)."""
import numpy as np
import pytest
import torch

from oracle import fbank as of


def _signals():
    rs = np.random.RandomState(0)
    t = np.arange(16000 * 2) / 16000.0
    noise = 0.05 * rs.randn(len(t))
    tone = 0.3 * np.sin(2 * np.pi * 1000.0 * t) + 0.01 * rs.randn(len(t)) + 0.02  # with a DC offset
    return {"noise": noise, "tone": tone, "short": noise[:400 + 160 * 3 + 57]}


def test_mel_banks_and_window_match_oracle_structure():
    from emoasr_amd import features as ft
    fb = ft.kaldi_mel_banks()
    assert fb.shape == (80, 257) and (fb >= 0).all() and fb[:, -1].max() == 0.0
    peak = fb.argmax(1)
    assert (np.diff(peak) >= 0).all() and peak[0] >= 1  # centres move up; bin 0 (DC) is below 20 Hz
    # a pure tone lands in the filter whose triangle covers its FFT bin
    x = _signals()["tone"] * 2 ** 15
    feats = of.fbank(x)
    assert feats.shape == (198, 80)
    tone_bin = int(round(1000.0 / 31.25))
    assert abs(int(feats.mean(0).argmax()) - int(fb[:, tone_bin].argmax())) <= 1
    w = ft.hamming_window(400)
    assert abs(w[0] - 0.08) < 1e-12 and abs(w[199] - w[200]) < 1e-12 and abs(w.max() - 1.0) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["noise", "tone", "short"])
def test_fbank_kernel_matches_oracle(dev, name):
    from emoasr_amd.features import LogMel
    x = _signals()[name]
    ref = of.fbank(x * 2 ** 15)
    lm = LogMel(dev)
    got = lm(torch.from_numpy(x).float().to(dev)).cpu().numpy()
    assert got.shape == ref.shape == (lm.num_frames(len(x)), 80)
    assert np.abs(got - ref).max() < 2e-3, np.abs(got - ref).max()  # f32 FFT / f32 input vs float64
