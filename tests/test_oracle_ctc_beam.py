"""Oracle part 4 (CTC prefix beam search with LM shallow fusion) pinned to the reference's outputs
(tests/golden/ctcbeam_tiny.npz, made by tests/golden/make_golden.py ctcbeam)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import ctc_beam as ob
from oracle import decoder as od
from oracle import model as om
from tests.util import CTC_BEAM_SETTINGS, LM_CFG, load_ctc_beam_golden, split_ragged


@pytest.mark.parametrize("si", range(len(CTC_BEAM_SETTINGS)))
def test_ctc_beam_search_matches_reference(si):
    cfg, sd, lmsd, g2, gb = load_ctc_beam_golden()
    st = CTC_BEAM_SETTINGS[si]
    lmcfg = SimpleNamespace(**LM_CFG)

    def lm_predict(batch, lens):
        with torch.no_grad():
            return od.lm_predict(lmsd, lmcfg, torch.from_numpy(batch), torch.from_numpy(lens)).numpy()

    for b in (1, 2, 3):
        n = int(g2["xlens"][b])
        with torch.no_grad():
            eouts, elens = om.encoder_forward(sd, cfg, g2["xs"][b:b + 1, :n], g2["xlens"][b:b + 1])
            logits = om.ctc_decoder_forward(sd, cfg, eouts, elens)
        if si == 0:
            ref_logits = gb[f"logits/{b}"]
            assert ((logits - ref_logits).abs().max() / ref_logits.abs().max()).item() < 1e-4
        lp = torch.log_softmax(logits[0].double(), -1).numpy()
        hyps, scores = ob.ctc_prefix_beam_search(lp, cfg.blank_id, cfg.eos_id, st["beam_width"], st["len_weight"],
                                                 lm_predict, st["lm_weight"])
        ref_hyps = split_ragged(gb[f"decode/{si}/{b}/hyps"], gb[f"decode/{si}/{b}/lens"])
        assert hyps == ref_hyps, (si, b)
        np.testing.assert_allclose(scores, gb[f"decode/{si}/{b}/scores"].numpy(), rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("lm_weight,len_weight,beam", [(0.0, 0.0, 4), (0.4, 0.3, 5), (0.3, 0.0, 10)])
def test_native_bookkeeping_equals_the_oracle(lm_weight, len_weight, beam):
    """csrc/ctc_beam_host.hip (emoasr_ctc_beam_step: the per-frame extend / merge / sort / prune in C, host code -- no GPU needed)
    against the oracle's search on random log-probabilities with a deterministic stand-in LM (a function of the prefix alone):
    the same hypotheses in the same order, scores to 1e-9 (the oracle folds with numpy.logaddexp, the C code with the reference's
    own max-factored form)."""
    import ctypes
    from emoasr_amd import lib
    L = lib.load()
    L.emoasr_ctc_beam_new.restype = ctypes.c_void_p
    L.emoasr_ctc_beam_new.argtypes = [ctypes.c_int] * 3 + [ctypes.c_double] * 2
    L.emoasr_ctc_beam_free.argtypes = [ctypes.c_void_p]
    L.emoasr_ctc_beam_free.restype = None
    L.emoasr_ctc_beam_step.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] + [ctypes.c_void_p] * 3
    L.emoasr_ctc_beam_scores.argtypes = [ctypes.c_void_p] * 2
    L.emoasr_ctc_beam_prefix.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    rng = np.random.RandomState(7)
    T, V, blank, eos = 80, 14, 0, 2
    x = (rng.randn(T, V) * 2.5).astype(np.float32)
    x[:, blank] += 1.0   # blanks win often, as in a trained model: prefixes stay and merge
    logp = (x - np.log(np.exp(x.astype(np.float64)).sum(1, keepdims=True))).astype(np.float32)

    def lm_row(prefix):   # deterministic "LM": log-soft-max of a pseudo-random row seeded by the prefix
        r = np.random.RandomState(hash(tuple(int(v) for v in prefix)) % (2 ** 31)).randn(V)
        return (r - np.log(np.exp(r).sum())).astype(np.float32)

    def lm_predict(batch, lens):
        return np.stack([lm_row(batch[i, : lens[i]]) for i in range(len(lens))])

    want_hyps, want_scores = ob.ctc_prefix_beam_search(logp.astype(np.float64), blank, eos, beam, len_weight,
                                                       lm_predict if lm_weight > 0 else None, lm_weight)
    k = min(beam, V)
    top = np.ascontiguousarray(np.argsort(-logp.astype(np.float64), 1, kind="stable")[:, :k].astype(np.int32))
    h = L.emoasr_ctc_beam_new(beam, blank, eos, len_weight, lm_weight)
    parent, tok = np.zeros(beam, np.int32), np.zeros(beam, np.int32)
    live = [(eos,)]
    for t in range(T):
        ptr = None
        if lm_weight > 0:
            vals = np.ascontiguousarray(np.stack([lm_row(p)[top[t]] for p in live]).astype(np.float64))
            ptr = vals.ctypes.data
        n = L.emoasr_ctc_beam_step(h, logp[t].ctypes.data, top[t].ctypes.data, k, ptr, parent.ctypes.data, tok.ctypes.data)
        assert n > 0
        live = [live[parent[i]] + ((int(tok[i]),) if tok[i] >= 0 else ()) for i in range(n)]
    scores = np.zeros(len(live))
    assert L.emoasr_ctc_beam_scores(h, scores.ctypes.data) == len(live)
    buf = np.zeros(T + 2, np.int32)
    for i, p in enumerate(live):   # the parent / label trail reproduces the prefixes the library holds
        n = L.emoasr_ctc_beam_prefix(h, i, buf.ctypes.data, len(buf))
        assert tuple(buf[:n].tolist()) == p
    L.emoasr_ctc_beam_free(h)
    assert [list(p) for p in live] == want_hyps
    np.testing.assert_allclose(scores, want_scores, rtol=1e-9, atol=1e-9)
