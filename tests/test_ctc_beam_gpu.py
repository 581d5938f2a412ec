"""CTC prefix beam search with LM shallow fusion on the GPU against the reference's outputs
(tests/golden/ctcbeam_tiny.npz): same hypotheses in the same order, scores within 1e-3."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from tests.util import CONFIGS, CTC_BEAM_SETTINGS, LM_CFG, load_ctc_beam_golden, split_ragged

pytestmark = pytest.mark.gpu


def _build(dtype, dev):
    from emoasr_amd.modeling.asr import ASR
    from emoasr_amd.modeling.lm import LM
    cfg, sd, lmsd, g2, gb = load_ctc_beam_golden()
    model = ASR(SimpleNamespace(**CONFIGS["l2_tiny"]), compute_dtype=dtype)
    model.load_state_dict(sd)
    lm = LM(SimpleNamespace(**LM_CFG), compute_dtype=dtype)
    lm.load_state_dict(lmsd)
    return model.to(dev).eval(), lm.to(dev).eval(), g2, gb


@pytest.mark.parametrize("si", range(len(CTC_BEAM_SETTINGS)))
def test_ctc_beam_search_f32(dev, si):
    model, lm, g2, gb = _build(torch.float32, dev)
    st = CTC_BEAM_SETTINGS[si]
    for b in (1, 2, 3):
        n = int(g2["xlens"][b])
        hyps, scores, logits, aligns = model.decode(g2["xs"][b:b + 1, :n].to(dev), g2["xlens"][b:b + 1], lm=lm, **st)
        assert aligns is None
        if si == 0:
            ref = gb[f"logits/{b}"]
            assert ((logits.float().cpu() - ref).abs().max() / ref.abs().max()).item() < 1e-3
        assert hyps == split_ragged(gb[f"decode/{si}/{b}/hyps"], gb[f"decode/{si}/{b}/lens"]), (si, b)
        np.testing.assert_allclose(scores, gb[f"decode/{si}/{b}/scores"].numpy(), rtol=1e-3, atol=1e-3)


def test_ctc_beam_search_bf16_runs(dev):
    """throughput mode: same code path end to end; the best hypothesis is a plausible label sequence"""
    model, lm, g2, gb = _build(torch.bfloat16, dev)
    n = int(g2["xlens"][3])
    hyps, scores, _, _ = model.decode(g2["xs"][3:4, :n].to(dev), g2["xlens"][3:4], beam_width=4, len_weight=0.1, lm=lm,
                                      lm_weight=0.3)
    assert len(hyps) == 4 and hyps[0][0] == 2 and all(0 < v < 40 for v in hyps[0])
    assert scores == sorted(scores, reverse=True) and all(np.isfinite(scores))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_lm_predict_graph_replay_equals_the_eager_call(dev, dtype, monkeypatch):
    """LM.predict for up to 16 rows replays its ~90 launches from a HIP graph over padded static buffers (rows to 4 / 16, length to
    a multiple of 8): bit-identical to the eager call for ragged prefixes, repeated calls and changing shapes"""
    _, lm, _, _ = _build(dtype, dev)
    g = torch.Generator().manual_seed(5)
    cases = [(5, 11, [11, 3, 7, 1, 9]), (2, 4, [4, 4]), (16, 9, list(range(1, 10)) + [9] * 7), (1, 1, [1]), (5, 11, [2, 11, 5, 5, 8])]
    for B, N, lens in cases:
        ys = torch.randint(3, 40, (B, N), generator=g)
        for b, n in enumerate(lens):
            ys[b, n:] = 0
        monkeypatch.setenv("EMOASR_LM_GRAPH", "0")
        want, _ = lm.predict(ys, lens)
        monkeypatch.setenv("EMOASR_LM_GRAPH", "1")
        got, _ = lm.predict(ys, lens)
        again, _ = lm.predict(ys, lens)
        assert torch.equal(got, want) and torch.equal(again, want), (B, N)


@pytest.mark.parametrize("si", range(len(CTC_BEAM_SETTINGS)))
def test_native_bookkeeping_equals_the_python_loop(dev, si, monkeypatch):
    """the prefix bookkeeping in C (csrc/ctc_beam_host.hip; LM rows cached on the device, a frame reads its k candidate columns)
    against the Python loop it replaces: the same hypotheses and BIT-identical float64 scores"""
    model, lm, g2, gb = _build(torch.float32, dev)
    st = CTC_BEAM_SETTINGS[si]
    for b in (1, 2, 3):
        n = int(g2["xlens"][b])
        xs, xl = g2["xs"][b:b + 1, :n].to(dev), g2["xlens"][b:b + 1]
        monkeypatch.setenv("EMOASR_CTC_BEAM_NATIVE", "0")
        want_h, want_s, _, _ = model.decode(xs, xl, lm=lm, **st)
        monkeypatch.setenv("EMOASR_CTC_BEAM_NATIVE", "1")
        got_h, got_s, _, _ = model.decode(xs, xl, lm=lm, **st)
        assert got_h == want_h, (si, b)
        assert list(got_s) == list(want_s), (si, b, got_s, want_s)
