"""Pin the CPU oracle (oracle/model.py) to golden vectors produced by the reference itself
(tests/golden/make_golden.py).  fp32 on both sides: tolerance 1e-4 relative."""
import pytest
import torch

from oracle import model as om
from tests.util import load_golden, split_ragged


def _rel(a, b):
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


@pytest.mark.parametrize("name", ["l2_tiny", "l1_tiny", "l2abs_tiny"])
def test_eval_forward_and_greedy(name):
    cfg, sd, g = load_golden(name)
    with torch.no_grad():
        eouts, elens = om.encoder_forward(sd, cfg, g["xs"], g["xlens"], training=False)
        loss, _, logits = om.ctc_decoder_forward(sd, cfg, eouts, elens, g["ys"], g["ylens"])
        hyps, aligns, _ = om.asr_ctc_greedy(sd, cfg, g["xs"], g["xlens"])
    assert torch.equal(elens, g["eval/elens"])
    assert _rel(eouts, g["eval/eouts"]) < 1e-4
    assert _rel(logits, g["eval/logits"]) < 1e-4
    assert abs(loss.item() - g["eval/loss"].item()) < 1e-3 * abs(g["eval/loss"].item())
    assert hyps == split_ragged(g["eval/hyps"], g["eval/hyp_lens"])  # bit-exact token ids
    assert sum(aligns, []) == g["eval/aligns"].tolist()


@pytest.mark.parametrize("name", ["l2_tiny", "l1_tiny", "l2abs_tiny"])
def test_train_loss_and_grads(name):
    cfg, sd, g = load_golden(name)
    sd = {k: v.clone() for k, v in sd.items()}
    params = {k: v for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    for v in params.values():
        v.requires_grad_(True)
    loss, _, _ = om.asr_ctc_forward(sd, cfg, g["xs"], g["xlens"], g["ys"], g["ylens"], training=True)
    loss.backward()
    assert abs(loss.item() - g["train/loss"].item()) < 1e-4 * abs(g["train/loss"].item())
    worst = 0.0
    gmax = max(g["grad/" + k].abs().max().item() for k in params)
    for k, v in params.items():
        ref = g["grad/" + k]
        # gradients that are analytically zero (bias before BatchNorm, key bias) are pure rounding noise
        worst = max(worst, ((v.grad - ref).abs().max() / max(ref.abs().max().item(), 1e-2 * gmax)).item())
    assert worst < 2e-3, worst
    for k in g:
        if k.startswith("sd_after/") and "running_" in k:
            assert _rel(sd[k[len("sd_after/"):]], g[k]) < 1e-5, k
