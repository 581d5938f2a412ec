"""Log-mel front end: the host constants against the oracle's independent restatement (CPU), the HIP kernel
against the oracle (GPU).  Parity with the reference's torchaudio call itself is UNPINNED (see oracle/fbank.py); the oracle is
cross-checked against an independent published implementation of the same recipe (transformers.audio_utils)."""
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


@pytest.mark.parametrize("name", ["noise", "tone", "short"])
def test_oracle_against_an_independent_kaldi_fbank(name):
    """CROSS-CHECK of oracle/fbank.py (the reference's own dependency, torchaudio.compliance.kaldi.fbank, is not installed and
    cannot be fetched): `transformers.audio_utils` carries an independent, published implementation of the same Kaldi recipe --
    the torchaudio-free path of the Hugging Face feature extractors (ASTFeatureExtractor._extract_fbank_features: `spectrogram`
    with remove_dc_offset, pre-emphasis 0.97, a Kaldi-scale mel bank triangularised in mel space, log floored at FLT_EPSILON) --
    here called with the reference's call-site options (hamming window, 80 bins from 20 Hz; wav_to_feats.py:26-33).  The two
    restatements agree to 1e-5 on log-mel values of magnitude ~30 (measured 1.0e-6)."""
    audio_utils = pytest.importorskip("transformers.audio_utils")
    x = _signals()[name] * 2 ** 15
    win = audio_utils.window_function(400, "hamming", periodic=False)
    mel = audio_utils.mel_filter_bank(num_frequency_bins=257, num_mel_filters=80, min_frequency=20, max_frequency=8000,
                                      sampling_rate=16000, norm=None, mel_scale="kaldi", triangularize_in_mel_space=True)
    other = audio_utils.spectrogram(x, win, frame_length=400, hop_length=160, fft_length=512, power=2.0, center=False,
                                    preemphasis=0.97, mel_filters=mel, log_mel="log", mel_floor=1.192092955078125e-07,
                                    remove_dc_offset=True).T
    ours = of.fbank(x)
    assert ours.shape == other.shape
    assert np.abs(ours - other).max() < 1e-5, np.abs(ours - other).max()


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
