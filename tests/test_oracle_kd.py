"""Oracle part 4 (knowledge distillation) pinned to the reference through tests/golden/kd_tiny.npz:
forced aligner, frame -> label mapping, the four distillation losses (values + logits gradients), and the
CTC / Transformer decoders with kd_weight > 0 on the l2_tiny / l3_tiny weights."""
from types import SimpleNamespace

import pytest
import torch

from oracle import distill as od
from oracle import model as om
from tests.util import (CAD_CASES, CONFIGS, DISTILL_CASES, KD_ATT, KD_CTC_CASES, KD_INTER_CASES, KD_RNNT_CASES, load_golden,
                        load_kd_golden)


def test_forced_aligner_and_label_map():
    g = load_kd_golden()
    lp = torch.log_softmax(g["align/logits"], -1)
    elens, ys, ylens = g["align/elens"], g["align/ys"], g["align/ylens"]
    aligns = od.ctc_forced_align(lp, elens, ys, ylens)
    assert torch.equal(aligns, g["align/aligns"])
    for pos in ("all", "left", "mid", "right"):
        for b in range(aligns.shape[0]):
            n = int(elens[b])
            assert od.frame_to_label_map(aligns[b, :n].tolist(), 0, pos) == g["align/map_" + pos][b, :n].tolist()
    # the reference's own smoke vector (criteria.py:290-298)
    assert od.frame_to_label_map([5, 0, 0, 15, 15, 15, 15, 10, 10, 0], 0, "all") == [0, -1, -1, 1, 1, 1, 1, 2, 2, -1]
    assert od.frame_to_label_map([5, 0, 0, 15, 15, 15, 15, 10, 10, 0], 0, "mid") == [0, -1, -1, -1, 1, -1, -1, 2, -1, -1]


def _check(loss, z, want, want_grad, tol=1e-5):
    loss.backward()
    assert abs(float(loss) - float(want)) < tol * max(1.0, abs(float(want)))
    assert (z.grad - want_grad).abs().max() < tol


def test_distillation_losses():
    g = load_kd_golden()
    elens, ys, ylens, soft = g["align/elens"], g["align/ys"], g["align/ylens"], g["loss/soft"]
    for name, kw in CAD_CASES.items():
        z = g["align/logits"].clone().requires_grad_(True)
        _check(od.ctc_align_distill_loss(z, ys, soft, g["align/aligns"], elens, ylens, **kw), z, g[f"loss/{name}"],
               g[f"loss/{name}_grad"])
    for name, kw in DISTILL_CASES.items():
        z = g["loss/dec_logits"].clone().requires_grad_(True)
        l, ls, lh = od.distill_loss(z, ys, soft, ylens, **kw)
        want = g[f"loss/{name}"]
        assert abs(float(ls) - float(want[1])) < 1e-4 and abs(float(lh) - float(want[2])) < 1e-4
        _check(l, z, want[0], g[f"loss/{name}_grad"])
    z = g["loss/rnnt_logits"].clone().requires_grad_(True)
    _check(od.rnnt_word_distill_loss(z, soft, g["loss/rnnt_xlens"], ylens), z, g["loss/rnnt_word"],
           g["loss/rnnt_word_grad"])
    z = g["loss/rnnt_logits"].clone().requires_grad_(True)
    _check(od.rnnt_align_distill_loss(z, ys, soft, g["loss/rnnt_aligns"], g["loss/rnnt_xlens"], ylens), z,
           g["loss/rnnt_align"], g["loss/rnnt_align_grad"])


def _grads(sd, build_loss):
    sd = {k: v.clone() for k, v in sd.items()}
    params = {k: v for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    for v in params.values():
        v.requires_grad_(True)
    loss, ld, extra = build_loss(sd)
    loss.backward()
    return loss, ld, extra, {k: v.grad for k, v in params.items() if v.grad is not None}


@pytest.mark.parametrize("case", list(KD_CTC_CASES))
def test_ctc_decoder_with_kd(case):
    _, sd, g2 = load_golden("l2_tiny")
    g = load_kd_golden()
    cfg = SimpleNamespace(**dict(CONFIGS["l2_tiny"], **KD_CTC_CASES[case]))

    def build(sd):
        eouts, elens = om.encoder_forward(sd, cfg, g2["xs"], g2["xlens"], training=True)
        loss, ld, _, aligns = od.ctc_decoder_forward_kd(sd, cfg, eouts, elens, g2["ys"], g2["ylens"], g["model/soft_ctc"])
        return loss, ld, aligns

    loss, ld, aligns, grads = _grads(sd, build)
    for k in ("loss_ctc", "loss_kd", "loss_total"):
        want = float(g[f"model/{case}/ld/{k}"])
        assert abs(float(ld[k]) - want) < 1e-4 * abs(want), (k, float(ld[k]), want)
    for k in [k for k in g if k.startswith(f"model/{case}/grad/")]:
        name = k.split("/grad/")[1]
        assert (grads[name] - g[k]).abs().max() < 1e-4 * max(1.0, g[k].abs().max().item()), name


def test_ctc_decoder_kd_aligns():
    _, sd, g2 = load_golden("l2_tiny")
    g = load_kd_golden()
    cfg = SimpleNamespace(**CONFIGS["l2_tiny"])
    with torch.no_grad():
        eouts, elens = om.encoder_forward(sd, cfg, g2["xs"], g2["xlens"], training=True)  # BatchNorm batch statistics
        logits = om.ctc_decoder_forward(sd, cfg, eouts, elens)
        aligns = od.ctc_forced_align(torch.log_softmax(logits, -1), elens, g2["ys"], g2["ylens"])
    assert torch.equal(aligns, g["model/ctc_all/aligns"])


def test_attention_decoder_with_kd():
    _, sd, g3 = load_golden("l3_tiny")
    g = load_kd_golden()
    cfg = SimpleNamespace(**dict(CONFIGS["l3_tiny"], **KD_ATT))

    def build(sd):
        eouts, elens = om.encoder_forward(sd, cfg, g3["xs"], g3["xlens"], training=True)
        loss, ld, _ = od.att_decoder_forward_kd(sd, cfg, eouts, elens, g3["ys"], g3["ylens"], g3["ys_in"], g3["ys_out"],
                                                g["model/soft_att"])
        return loss, ld, None

    loss, ld, _, grads = _grads(sd, build)
    for k in ("loss_kd", "loss_att", "loss_ctc", "loss_total"):
        want = float(g[f"model/att/ld/{k}"])
        assert abs(float(ld[k]) - want) < 1e-4 * abs(want), (k, float(ld[k]), want)
    for k in [k for k in g if k.startswith("model/att/grad/")]:
        name = k.split("/grad/")[1]
        assert (grads[name] - g[k]).abs().max() < 1e-4 * max(1.0, g[k].abs().max().item()), name


@pytest.mark.parametrize("case", list(KD_INTER_CASES))
def test_ctc_decoder_intermediate_and_phone_branches(case):
    _, sd, g2 = load_golden("l2_tiny")
    g = load_kd_golden()
    cfg = SimpleNamespace(**dict(CONFIGS["l2_tiny"], **KD_INTER_CASES[case]))
    sd = dict(sd, **{k.split("/sd/")[1]: v for k, v in g.items() if k.startswith(f"model/{case}/sd/")})

    def build(sd):
        eouts, elens, inter = od.encoder_forward_inter(sd, cfg, g2["xs"], g2["xlens"], training=True)
        loss, ld, _ = od.ctc_decoder_forward_full(sd, cfg, eouts, elens, inter, g2["ys"], g2["ylens"],
                                                  g["model/soft_ctc"], g["model/ps"], g["model/plens"])
        return loss, ld, None

    loss, ld, _, grads = _grads(sd, build)
    want_keys = [k.split("/ld/")[1] for k in g if k.startswith(f"model/{case}/ld/")]
    assert sorted(ld) == sorted(want_keys)
    for k in want_keys:
        want = float(g[f"model/{case}/ld/{k}"])
        assert abs(float(ld[k]) - want) < 1e-4 * abs(want), (k, float(ld[k]), want)
    for k in [k for k in g if k.startswith(f"model/{case}/grad/")]:
        name = k.split("/grad/")[1]
        assert (grads[name] - g[k]).abs().max() < 1e-4 * max(1.0, g[k].abs().max().item()), name


@pytest.mark.parametrize("case", list(KD_RNNT_CASES))
def test_rnnt_decoder_with_word_kd(case):
    _, sd, g4 = load_golden("l4_tiny")
    g = load_kd_golden()
    cfg = SimpleNamespace(**dict(CONFIGS["l4_tiny"], **KD_RNNT_CASES[case]))

    def build(sd):
        eouts, elens = om.encoder_forward(sd, cfg, g4["xs"], g4["xlens"], training=True)
        loss, ld, _ = od.rnnt_decoder_forward_kd(sd, cfg, eouts, elens, g4["ys"], g4["ylens"], g4["ys_in"],
                                                 g["model/soft_rnnt"])
        return loss, ld, None

    loss, ld, _, grads = _grads(sd, build)
    for k in ("loss_rnnt", "loss_ctc", "loss_kd", "loss_total"):
        want = float(g[f"model/{case}/ld/{k}"])
        assert abs(float(ld[k]) - want) < 1e-4 * abs(want), (k, float(ld[k]), want)
    for k in [k for k in g if k.startswith(f"model/{case}/grad/")]:
        name = k.split("/grad/")[1]
        assert (grads[name] - g[k]).abs().max() < 1e-4 * max(1.0, g[k].abs().max().item()), name


def test_rnnt_forced_align_walk():
    """the lattice part is unpinned (numba absent); the walk is checked on a case decidable by hand:
    a sharply peaked posterior must reproduce the planted alignment, and labels left over when the last
    frame is reached keep frame 0 (rnnt_aligner.py:186-196)."""
    T, U, V = 6, 3, 5
    labels = torch.tensor([[1, 2, 3]])
    emit_at = [1, 1, 4]  # frame at which each label is emitted
    z = torch.full((1, T, U + 1, V), -8.0)
    u = 0
    for t in range(T):
        while u < U and emit_at[u] == t:
            z[0, t, u, labels[0, u]] = 8.0
            u += 1
        z[0, t, u, 0] = 8.0
    lp = torch.log_softmax(z, -1)
    assert od.rnnt_forced_align(lp, [T], labels, [U]).tolist() == [emit_at]
    # labels forced onto the last frame are never written by the walk
    z2 = torch.full((1, T, U + 1, V), -8.0)
    z2[0, :, 0, 0] = 8.0
    z2[0, T - 1, 0, 1] = 9.0
    z2[0, T - 1, 1, 2] = 9.0
    z2[0, T - 1, 2, 3] = 9.0
    z2[0, T - 1, 3, 0] = 9.0
    assert od.rnnt_forced_align(torch.log_softmax(z2, -1), [T], labels, [U]).tolist() == [[0, 0, 0]]
