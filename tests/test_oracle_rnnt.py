"""Oracle part 3 (RNN-T): the transducer loss restatement is validated by brute-force path
enumeration and finite differences (warp_rnnt is absent: parity unpinned for the loss value);
the network around it is pinned to the reference through tests/golden/l4_tiny.npz."""
import itertools

import numpy as np
import pytest
import torch

from oracle import model as om
from oracle import rnnt as orn
from tests.util import load_golden, split_ragged


def _brute(lp, labels, T, U, blank):
    """sum over all monotone paths: T blanks (one per frame) interleaved with U labels in order"""
    total = 0.0
    for pos in itertools.combinations(range(T + U), U):  # positions of label emissions
        t = u = 0
        logp = 0.0
        ok = True
        for step in range(T + U):
            if step in pos:
                logp += lp[t, u, labels[u]].item()
                u += 1
            else:
                logp += lp[t, u, blank].item()
                t += 1
            if t == T and step < T + U - 1:
                ok = False  # the last blank must be the final step
                break
        if ok and t == T and u == U:
            total += np.exp(logp)
    return -np.log(total)


def test_loss_against_path_enumeration():
    torch.manual_seed(0)
    for T, U in [(1, 0), (3, 2), (4, 3), (2, 3)]:
        V = 5
        lp = torch.log_softmax(torch.randn(1, T, U + 1, V, dtype=torch.float64), -1)
        labels = torch.randint(1, V, (1, max(U, 1)))
        got = orn.rnnt_nll(lp, labels, [T], [U], blank=0)[0].item()
        assert abs(got - _brute(lp[0], labels[0], T, U, 0)) < 1e-9, (T, U)


def test_loss_gradient_finite_differences():
    torch.manual_seed(1)
    T, U, V = 4, 2, 4
    z = torch.randn(2, T, U + 1, V, dtype=torch.float64, requires_grad=True)
    labels = torch.randint(1, V, (2, U))
    flens, llens = [4, 3], [2, 1]
    f = lambda x: orn.rnnt_loss(torch.log_softmax(x, -1), labels, flens, llens, reduction="mean", blank=0)
    assert torch.autograd.gradcheck(f, (z,), eps=1e-6, atol=1e-6)
    # cells outside (flens, llens) get no gradient
    f(z).backward()
    assert z.grad[1, 3].abs().max() == 0 and z.grad[1, :, 2].abs().max() == 0


def test_network_pinned_to_reference():
    cfg, sd, g = load_golden("l4_tiny")
    with torch.no_grad():
        eouts, elens = om.encoder_forward(sd, cfg, g["xs"], g["xlens"])
        douts, _ = orn.recurrency(sd, cfg, g["ys_in"])
        jl = orn.joint(sd, eouts[:1, :20], douts[:1])
        hyps, aligns = orn.rnnt_greedy(sd, cfg, eouts, elens)
    rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
    assert rel(douts, g["eval/douts"]) < 1e-4
    assert rel(jl, g["eval/joint_logits_b0"]) < 1e-4
    assert hyps == split_ragged(g["eval/hyps"], g["eval/hyp_lens"])
    assert aligns == split_ragged(g["eval/aligns"], g["eval/align_lens"])


def test_train_loss_and_grads():
    cfg, sd, g = load_golden("l4_tiny")
    sd = {k: v.clone() for k, v in sd.items()}
    params = {k: v for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    for v in params.values():
        v.requires_grad_(True)
    eouts, elens = om.encoder_forward(sd, cfg, g["xs"], g["xlens"], training=True)
    loss, ld, _ = orn.rnnt_decoder_forward(sd, cfg, eouts, elens, g["ys"], g["ylens"], g["ys_in"])
    loss.backward()
    for k, ref in (("loss_rnnt", "train/loss_rnnt"), ("loss_ctc", "train/loss_ctc"), ("loss_total", "train/loss")):
        assert abs(ld[k].item() - g[ref].item()) < 1e-4 * abs(g[ref].item()), k
    gmax = max(g["grad/" + k].abs().max().item() for k in params)
    worst = max(((v.grad - g["grad/" + k]).abs().max() / max(g["grad/" + k].abs().max().item(), 1e-2 * gmax)).item()
                for k, v in params.items())
    assert worst < 5e-3, worst


def test_beam_search_pinned_to_reference():
    """ALSD beam search (rnn_transducer.py:242-325): every surviving hypothesis, in order"""
    from tests.util import RNNT_BEAM_WIDTHS, load_rnnt_beam_golden
    cfg, sd, g = load_golden("l4_tiny")
    want = load_rnnt_beam_golden()
    with torch.no_grad():
        for b in range(g["xs"].shape[0]):
            n = int(g["xlens"][b])
            eouts, _ = om.encoder_forward(sd, cfg, g["xs"][b:b + 1, :n], g["xlens"][b:b + 1])
            for bw in RNNT_BEAM_WIDTHS:
                assert orn.rnnt_beam_search(sd, cfg, eouts, bw) == want[bw][b], (b, bw)


@pytest.mark.parametrize("fixture", ["rnnt_align_xcheck.npz", "rnnt_align_xcheck2.npz"])
def test_lattice_against_the_reference_aligner_kernels(fixture):
    """CROSS-CHECK (does not pin the loss value: warp_rnnt stays absent).  asr/modeling/decoders/rnnt_aligner.py:14-152 is the
    reference's own statement of the transducer forward / backward recursions; tests/golden/make_golden.py
    (run_rnnt_align_xcheck) executed those two kernel bodies as plain Python and stored alpha, beta, log_p and the alignments
    of RNNTForcedAligner.__call__ (:158-198).  The oracle's lattice (oracle.distill.rnnt_alpha_beta), its loss
    (oracle.rnnt.rnnt_nll: log_p = -nll / T) and its alignment walk must reproduce them."""
    import os

    from oracle import distill as od
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", fixture))   # (the second set: 6 ragged lattices up to 23 x 12 x 13)
    lp, ys = torch.from_numpy(z["log_probs"]), torch.from_numpy(z["ys"])
    elens, ylens = torch.from_numpy(z["elens"]), torch.from_numpy(z["ylens"])
    for b in range(lp.shape[0]):
        T, U = int(elens[b]), int(ylens[b])
        alpha, beta = od.rnnt_alpha_beta(lp[b].double(), [int(v) for v in ys[b, :U]], T, U, 0)
        assert np.allclose(alpha.numpy(), z["alpha"][b, :T, :U + 1], atol=1e-5), b
        assert np.allclose(beta.numpy(), z["beta"][b, :T, :U + 1], atol=1e-5), b
    nll = orn.rnnt_nll(lp, ys, elens, ylens, 0)
    assert np.allclose((-nll / elens).numpy(), z["log_p_alpha"], atol=1e-5)
    assert np.allclose((-nll / elens).numpy(), z["log_p_beta"], atol=1e-5)
    assert torch.equal(od.rnnt_forced_align(lp, elens, ys, ylens, 0), torch.from_numpy(z["aligns"]))
