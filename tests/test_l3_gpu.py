"""Transformer-decoder (L3) parity on the GPU against the reference goldens (l3_tiny)."""
from types import SimpleNamespace

import pytest
import torch

from tests.util import CONFIGS, DECODE_SETTINGS, LM_CFG, load_golden, lm_state, split_ragged

pytestmark = pytest.mark.gpu


def _build(dtype, dev):
    from emoasr_amd.modeling.asr import ASR
    cfg, sd, g = load_golden("l3_tiny")
    model = ASR(SimpleNamespace(**CONFIGS["l3_tiny"]), compute_dtype=dtype)
    model.load_state_dict(sd)
    return model.to(dev), g


def _rel(a, b):
    return ((a.float().cpu() - b).abs().max() / (b.abs().max() + 1e-12)).item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_teacher_forced_logits(dev, dtype):
    model, g = _build(dtype, dev)
    model.eval()
    with torch.no_grad():
        eouts, elens, _ = model.encoder(g["xs"].to(dev), g["xlens"])
        logits = model.decoder(eouts, elens, None, g["ys"], g["ylens"], g["ys_in"], None)
    tol = 1e-3 if dtype == torch.float32 else 6e-2
    assert _rel(logits, g["eval/att_logits"]) < tol, _rel(logits, g["eval/att_logits"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_train_loss_and_grads(dev, dtype):
    model, g = _build(dtype, dev)
    model.train()
    loss, ld = model(g["xs"].to(dev), g["xlens"], g["ys"], g["ylens"], g["ys_in"], g["ys_out"])
    assert set(ld) == {"loss_att", "loss_ctc", "loss_total"}
    loss.backward()
    ltol = 1e-3 if dtype == torch.float32 else 2e-2
    for k, ref in (("loss_att", "train/loss_att"), ("loss_ctc", "train/loss_ctc"), ("loss_total", "train/loss")):
        assert abs(ld[k].item() - g[ref].item()) < ltol * abs(g[ref].item()), (k, ld[k].item(), g[ref].item())
    gmax = max(g[k].abs().max().item() for k in g if k.startswith("grad/"))
    worst, worst_name, cos_min, cos_name = 0.0, None, 1.0, None
    for n, p in model.named_parameters():
        ref = g["grad/" + n]
        got = p.grad.float().cpu()
        assert torch.isfinite(got).all(), n
        err = ((got - ref).abs().max() / max(ref.abs().max().item(), 1e-2 * gmax)).item()
        if err > worst:
            worst, worst_name = err, n
        if ref.abs().max() > 1e-2 * gmax:
            cos = torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0).item()
            if cos < cos_min:
                cos_min, cos_name = cos, n
    if dtype == torch.float32:
        assert worst < 5e-3, (worst, worst_name)
    else:
        assert cos_min > 0.98, (cos_min, cos_name, worst, worst_name)


def test_label_smoothing_module(dev):
    from emoasr_amd.criteria import LabelSmoothingLoss
    from oracle.decoder import label_smoothing_loss
    torch.manual_seed(0)
    B, L, V = 3, 7, 50
    logits = torch.randn(B, L, V)
    ys = torch.randint(0, V, (B, L))
    ylens = torch.tensor([7, 4, 1])
    for nl, nb in ((False, True), (True, True), (False, False)):
        ref_in = logits.clone().requires_grad_(True)
        ref = label_smoothing_loss(ref_in, ys, ylens, V, 0.1, nl, nb)
        ref.backward()
        x = logits.to(dev).requires_grad_(True)
        got = LabelSmoothingLoss(V, 0.1, nl, nb)(x, ys, ylens)
        (got * 2.0).backward()
        assert abs(got.item() - ref.item()) < 1e-4 * abs(ref.item())
        assert torch.allclose(x.grad.cpu(), 2.0 * ref_in.grad, rtol=1e-4, atol=1e-6)
