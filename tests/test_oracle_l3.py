"""Oracle part 2 (Transformer decoder, label smoothing, prefix scorer, joint beam search, LM)
pinned to golden vectors produced by the reference (tests/golden/make_golden.py: l3_tiny)."""
from types import SimpleNamespace

import numpy as np
import torch

from oracle import decoder as od
from oracle import model as om
from tests.util import DECODE_SETTINGS, LM_CFG, load_golden, lm_state, split_ragged


def _rel(a, b):
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def test_decoder_logits_and_lm():
    cfg, sd, g = load_golden("l3_tiny")
    with torch.no_grad():
        eouts, elens = om.encoder_forward(sd, cfg, g["xs"], g["xlens"])
        logits = od.decoder_logits(sd, cfg, eouts, elens, g["ys_in"], g["ylens"] + 1)
        lp = od.lm_predict(lm_state(g), SimpleNamespace(**LM_CFG), g["lm_test/ys"], g["lm_test/ylens"])
    assert _rel(eouts, g["eval/eouts"]) < 1e-4
    assert _rel(logits, g["eval/att_logits"]) < 1e-4
    assert _rel(lp, g["lm_test/logp"]) < 1e-4


def test_train_loss_and_grads():
    cfg, sd, g = load_golden("l3_tiny")
    sd = {k: v.clone() for k, v in sd.items()}
    params = {k: v for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    for v in params.values():
        v.requires_grad_(True)
    xs = g["xs"][:, : int(g["xlens"].max())]
    eouts, elens = om.encoder_forward(sd, cfg, xs, g["xlens"], training=True)
    loss, ld, _ = od.decoder_forward(sd, cfg, eouts, elens, g["ys"], g["ylens"], g["ys_in"], g["ys_out"])
    loss.backward()
    for k, ref in (("loss_att", "train/loss_att"), ("loss_ctc", "train/loss_ctc"), ("loss_total", "train/loss")):
        assert abs(ld[k].item() - g[ref].item()) < 1e-4 * abs(g[ref].item()), k
    gmax = max(g["grad/" + k].abs().max().item() for k in params)
    worst = max(((v.grad - g["grad/" + k]).abs().max() / max(g["grad/" + k].abs().max().item(), 1e-2 * gmax)).item()
                for k, v in params.items())
    assert worst < 5e-3, worst


def test_joint_beam_search_matches_reference():
    cfg, sd, g = load_golden("l3_tiny")
    lm = (lm_state(g), SimpleNamespace(**LM_CFG))
    for si, st in enumerate(DECODE_SETTINGS):
        for b in range(2):
            n = int(g["xlens"][b])
            with torch.no_grad():
                eouts, elens = om.encoder_forward(sd, cfg, g["xs"][b:b + 1, :n], g["xlens"][b:b + 1])
                hyps, scores = od.joint_beam_search(sd, cfg, eouts, elens, st["beam_width"], st["len_weight"], lm,
                                                    st["lm_weight"], st["decode_ctc_weight"])
            want = split_ragged(g[f"decode/{si}/{b}/hyps"], g[f"decode/{si}/{b}/lens"])
            assert hyps == want, (si, b, hyps, want)
            assert np.allclose(scores, g[f"decode/{si}/{b}/scores"].numpy(), rtol=1e-4, atol=1e-3), (si, b)


def test_prefix_scorer_brute_force():
    """CTC prefix score = log of the summed probability of every labelling that starts with the
    prefix, checked by enumerating all frame paths on a tiny lattice."""
    import itertools
    rs = np.random.RandomState(0)
    T, V, blank, eos = 5, 4, 0, 3
    x = np.log(rs.dirichlet(np.ones(V), size=T)).astype(np.float32)

    def collapse(path):
        out, prev = [], -1
        for p in path:
            if p != prev and p != blank:
                out.append(p)
            prev = p
        return out

    def prefix_prob(prefix):
        tot = 0.0
        for path in itertools.product(range(V), repeat=T):
            lab = collapse(path)
            if lab[: len(prefix)] == prefix:
                tot += np.exp(sum(x[t, p] for t, p in enumerate(path)))
        return tot

    sc = od.CTCPrefixScorer(x, blank, eos)
    r0 = sc.initial_state()
    cands = np.array([1, 2])
    psi, states = sc([eos], cands, r0)
    for c, p in zip(cands, psi):
        assert abs(np.exp(p) - prefix_prob([c])) < 1e-5
    psi2, _ = sc([eos, 1], cands, states[0])
    for c, p in zip(cands, psi2):
        assert abs(np.exp(p) - prefix_prob([1, c])) < 1e-5
