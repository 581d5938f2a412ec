"""Knowledge-distillation path and the auxiliary CTC branches on the GPU, against
tests/golden/kd_tiny.npz produced by the reference: forced aligner, frame -> label mapping, the four
distillation losses (values + logits gradients), and the CTC / Transformer decoders with kd_weight > 0,
intermediate CTC (+ KD) and phone-level CTC on the l2_tiny / l3_tiny weights.

Tolerances: f32 mode 1e-3 relative on losses, integer outputs (alignments, label maps) bit-exact;
bf16 mode 3e-2 on losses, gradients by cosine similarity."""
from types import SimpleNamespace

import pytest
import torch

from tests.util import (CAD_CASES, CONFIGS, DISTILL_CASES, KD_ATT, KD_CTC_CASES, KD_INTER_CASES, KD_RNNT_CASES, load_golden,
                        load_kd_golden)

pytestmark = pytest.mark.gpu


def test_forced_aligner_and_label_map(dev):
    from emoasr_amd import ops
    from emoasr_amd.modeling.decoders.ctc_aligner import CTCForcedAligner
    g = load_kd_golden()
    lp = torch.log_softmax(g["align/logits"], -1).to(dev)
    keep = lp.clone()
    aligns = CTCForcedAligner(blank_id=0)(lp, g["align/elens"], g["align/ys"], g["align/ylens"])
    assert aligns.dtype == torch.int64 and torch.equal(aligns.cpu(), g["align/aligns"])
    assert torch.equal(lp, keep)  # the argument is left alone
    xl = g["align/elens"].to(torch.int32).to(dev)
    for pos in ("all", "left", "mid", "right"):
        lmap, count = ops.ctc_label_map(aligns.to(torch.int32), xl, 0, pos)
        want = g["align/map_" + pos]
        for b in range(want.shape[0]):
            n = int(g["align/elens"][b])
            assert lmap[b, :n].cpu().tolist() == want[b, :n].tolist(), (pos, b)
            assert (lmap[b, n:] == -1).all()
            assert int(count[b]) == int((want[b, :n] >= 0).sum())


def test_label_map_long_utterance(dev):
    """more frames than one 256-wide scan chunk; every position mode against the oracle"""
    from emoasr_amd import ops
    from oracle import distill as od
    gen = torch.Generator().manual_seed(0)
    T = 700
    rows, lens = [], [700, 613, 257, 256, 1]
    for n in lens:
        toks = torch.randint(0, 4, (T,), generator=gen)
        toks = toks.repeat_interleave(torch.randint(1, 6, (T,), generator=gen))[:T]
        rows.append(toks)
    al = torch.stack(rows).to(torch.int32)
    for pos in ("all", "left", "mid", "right"):
        lmap, count = ops.ctc_label_map(al.to(dev), torch.tensor(lens, dtype=torch.int32, device=dev), 0, pos)
        for b, n in enumerate(lens):
            want = od.frame_to_label_map(al[b, :n].tolist(), 0, pos)
            assert lmap[b, :n].cpu().tolist() == want, (pos, b)
            assert int(count[b]) == sum(1 for v in want if v >= 0)


def _check(loss, z, want, want_grad, dtype):
    loss.backward()
    ltol, gtol = (1e-4, 1e-5) if dtype == torch.float32 else (2e-2, 2e-2)
    assert abs(loss.item() - float(want)) < ltol * max(1.0, abs(float(want))), (loss.item(), float(want))
    err = (z.grad.float().cpu() - want_grad).abs().max().item()
    assert err < gtol * max(1.0, want_grad.abs().max().item()), err


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_distillation_losses(dev, dtype):
    from emoasr_amd.criteria import CTCAlignDistillLoss, DistillLoss, RNNTAlignDistillLoss, RNNTWordDistillLoss
    g = load_kd_golden()
    elens, ys, ylens, soft = g["align/elens"], g["align/ys"], g["align/ylens"], g["loss/soft"]
    V = soft.shape[-1]
    leaf = lambda t: t.to(dtype).to(dev).clone().requires_grad_(True)
    ref = lambda t: t.to(dtype).float().clone()  # what the kernel actually sees in bf16 mode
    from oracle import distill as od
    for name, kw in CAD_CASES.items():
        z = leaf(g["align/logits"])
        loss = CTCAlignDistillLoss(vocab_size=V, blank_id=0, **kw)(z, ys, soft.to(dev), g["align/aligns"].to(dev), elens, ylens)
        if dtype == torch.float32:
            _check(loss, z, g[f"loss/{name}"], g[f"loss/{name}_grad"], dtype)
        else:
            zr = ref(g["align/logits"]).requires_grad_(True)
            lr = od.ctc_align_distill_loss(zr, ys, soft, g["align/aligns"], elens, ylens, **kw)
            lr.backward()
            _check(loss, z, lr.detach(), zr.grad, dtype)
    for name, kw in DISTILL_CASES.items():
        z = leaf(g["loss/dec_logits"])
        l, ls, lh = DistillLoss(vocab_size=V, **kw)(z, ys, soft.to(dev), ylens)
        want = g[f"loss/{name}"]
        if dtype == torch.float32:
            assert abs(ls.item() - float(want[1])) < 1e-4 * abs(float(want[1]))
            assert abs(lh.item() - float(want[2])) < 1e-4 * abs(float(want[2]))
            _check(l, z, want[0], g[f"loss/{name}_grad"], dtype)
        else:
            zr = ref(g["loss/dec_logits"]).requires_grad_(True)
            lr, _, _ = od.distill_loss(zr, ys, soft, ylens, **kw)
            lr.backward()
            _check(l, z, lr.detach(), zr.grad, dtype)
    xl4 = g["loss/rnnt_xlens"]
    z = leaf(g["loss/rnnt_logits"])
    l = RNNTWordDistillLoss()(z, soft.to(dev), xl4, ylens)
    zr = ref(g["loss/rnnt_logits"]).requires_grad_(True)
    lr = od.rnnt_word_distill_loss(zr, soft, xl4, ylens)
    lr.backward()
    if dtype == torch.float32:
        assert abs(lr.item() - g["loss/rnnt_word"].item()) < 1e-5
    _check(l, z, lr.detach(), zr.grad, dtype)
    z = leaf(g["loss/rnnt_logits"])
    l = RNNTAlignDistillLoss()(z, ys, soft.to(dev), g["loss/rnnt_aligns"], xl4, ylens)
    zr = ref(g["loss/rnnt_logits"]).requires_grad_(True)
    lr = od.rnnt_align_distill_loss(zr, ys, soft, g["loss/rnnt_aligns"], xl4, ylens)
    lr.backward()
    if dtype == torch.float32:
        assert abs(lr.item() - g["loss/rnnt_align"].item()) < 1e-5
        assert (zr.grad - g["loss/rnnt_align_grad"]).abs().max() < 1e-6
    _check(l, z, lr.detach(), zr.grad, dtype)


def _model(base, extra, dtype, dev, g, case):
    from emoasr_amd.modeling.asr import ASR
    _, sd, gb = load_golden(base)
    sd = {k: v for k, v in sd.items() if not k.startswith("lm.")}
    sd.update({k.split("/sd/")[1]: v for k, v in g.items() if k.startswith(f"model/{case}/sd/")})
    model = ASR(SimpleNamespace(**dict(CONFIGS[base], **extra)), compute_dtype=dtype)
    model.load_state_dict(sd)
    return model.to(dev), gb


def _compare(model, loss, ld, g, case, dtype):
    want_keys = [k.split("/ld/")[1] for k in g if k.startswith(f"model/{case}/ld/")]
    assert sorted(ld) == sorted(want_keys), (sorted(ld), sorted(want_keys))
    ltol = 1e-3 if dtype == torch.float32 else 3e-2
    for k in want_keys:
        want = float(g[f"model/{case}/ld/{k}"])
        assert abs(ld[k].item() - want) < ltol * abs(want), (k, ld[k].item(), want)
    loss.backward()
    grads = {n: p.grad.float().cpu() for n, p in model.named_parameters()}
    for k in [k for k in g if k.startswith(f"model/{case}/grad/")]:
        name = k.split("/grad/")[1]
        a, b = grads[name].flatten(), g[k].flatten()
        cos = torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)
        if dtype == torch.float32:
            assert (a - b).abs().max() < 2e-3 * max(1e-3, b.abs().max().item()), (name, (a - b).abs().max().item())
        assert cos > (0.9999 if dtype == torch.float32 else 0.98), (name, cos.item())
    norms = torch.tensor([grads[n].norm().item() for n, _ in model.named_parameters()])
    want = g[f"model/{case}/gnorms"]
    assert norms.shape == want.shape
    # (parameters whose true gradient is ~0, e.g. the key bias, only have rounding noise: floor the denominator)
    floor = 1e-4 if dtype == torch.float32 else 2e-2
    relv = (norms - want).abs() / (want + floor * want.max())
    rel, worst = relv.max().item(), [n for n, _ in model.named_parameters()][int(relv.argmax())]
    # (f32: 4.9e-3 / 5.3e-3 measured over two library builds, the worst parameter's gradient ~1e-4 of the largest norm)
    assert rel < (5e-3 if dtype == torch.float32 else 0.25), (rel, worst, norms[int(relv.argmax())].item(), want.max().item())


@pytest.mark.parametrize("case", list(KD_CTC_CASES) + list(KD_INTER_CASES))
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_ctc_decoder_branches(dev, dtype, case):
    g = load_kd_golden()
    extra = dict(KD_CTC_CASES, **KD_INTER_CASES)[case]
    model, g2 = _model("l2_tiny", extra, dtype, dev, g, case)
    model.train()
    loss, ld = model(g2["xs"].to(dev), g2["xlens"], g2["ys"], g2["ylens"], g2["ys_in"], g2["ys_out"],
                     soft_labels=g["model/soft_ctc"].to(dev), ps=g["model/ps"], plens=g["model/plens"])
    _compare(model, loss, ld, g, case, dtype)


def test_ctc_decoder_kd_aligns(dev):
    g = load_kd_golden()
    model, g2 = _model("l2_tiny", KD_CTC_CASES["ctc_all"], torch.float32, dev, g, "ctc_all")
    model.train()
    with torch.no_grad():
        eouts, elens, _ = model.encoder(g2["xs"].to(dev), g2["xlens"])
        logits = model.decoder(eouts, elens)
        aligns = model.decoder.forced_aligner(torch.log_softmax(logits.float(), -1), elens, g2["ys"], g2["ylens"])
    assert torch.equal(aligns.cpu(), g["model/ctc_all/aligns"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_attention_decoder_with_kd(dev, dtype):
    g = load_kd_golden()
    model, g3 = _model("l3_tiny", KD_ATT, dtype, dev, g, "att")
    model.train()
    loss, ld = model(g3["xs"].to(dev), g3["xlens"], g3["ys"], g3["ylens"], g3["ys_in"], g3["ys_out"],
                     soft_labels=g["model/soft_att"].to(dev))
    _compare(model, loss, ld, g, "att", dtype)


@pytest.mark.parametrize("case", list(KD_RNNT_CASES))
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_rnnt_decoder_with_word_kd(dev, dtype, case):
    g = load_kd_golden()
    model, g4 = _model("l4_tiny", KD_RNNT_CASES[case], dtype, dev, g, case)
    model.train()
    loss, ld = model(g4["xs"].to(dev), g4["xlens"], g4["ys"], g4["ylens"], g4["ys_in"], g4["ys_out"],
                     soft_labels=g["model/soft_rnnt"].to(dev))
    _compare(model, loss, ld, g, case, dtype)


def test_rnnt_forced_aligner(dev):
    """HIP lattice + walk against the oracle restatement (the reference's Numba kernels cannot run here)"""
    from emoasr_amd.modeling.decoders.rnnt_aligner import RNNTForcedAligner
    from oracle import distill as od
    gen = torch.Generator().manual_seed(4)
    B, T, L, V = 5, 19, 6, 9
    lp = torch.log_softmax(3.0 * torch.randn(B, T, L + 1, V, generator=gen), -1)
    ys = torch.randint(1, V, (B, L), generator=gen)
    elens, ylens = torch.tensor([19, 15, 9, 2, 1]), torch.tensor([6, 4, 6, 3, 2])
    got = RNNTForcedAligner(blank_id=0)(lp.to(dev), elens, ys, ylens)
    want = od.rnnt_forced_align(lp, elens, ys, ylens)
    assert got.dtype == torch.int32 and torch.equal(got.cpu(), want)


@pytest.mark.parametrize("fixture", ["rnnt_align_xcheck.npz", "rnnt_align_xcheck2.npz"])
def test_rnnt_lattice_against_the_reference_aligner_kernels(dev, fixture):
    """the HIP transducer lattice (emoasr_rnnt_forward: alpha, beta, nll) and RNNTForcedAligner against what the reference's OWN
    recursion bodies gave (rnnt_aligner.py:14-198 executed as plain Python by make_golden.py: rnnt_align_xcheck.npz) -- a
    cross-check of the lattice the transducer loss runs on; the loss value's third-party source (warp_rnnt) stays unpinned"""
    import os

    import numpy as np

    from emoasr_amd import ops
    from emoasr_amd.modeling.decoders.rnnt_aligner import RNNTForcedAligner
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", fixture))   # (the second set: 6 ragged lattices up to 23 x 12 x 13)
    lp, ys = torch.from_numpy(z["log_probs"]), torch.from_numpy(z["ys"])
    elens, ylens = torch.from_numpy(z["elens"]), torch.from_numpy(z["ylens"])
    got = RNNTForcedAligner(blank_id=0)(lp.to(dev), elens, ys, ylens)
    assert torch.equal(got.cpu(), torch.from_numpy(z["aligns"]))
    with ops.stream_scope():
        (_, _, _, alpha, beta), nll = ops.rnnt_forward(lp.to(dev).contiguous(), ys.to(torch.int32).to(dev),
                                                       elens.to(torch.int32).to(dev), ylens.to(torch.int32).to(dev), 0)
    alpha, beta, nll = alpha.cpu().numpy(), beta.cpu().numpy(), nll.cpu()
    for b in range(lp.shape[0]):
        T, U = int(elens[b]), int(ylens[b])
        assert np.allclose(alpha[b, :T, :U + 1], z["alpha"][b, :T, :U + 1], atol=1e-4), b
        assert np.allclose(beta[b, :T, :U + 1], z["beta"][b, :T, :U + 1], atol=1e-4), b
    assert np.allclose((-nll / elens).numpy(), z["log_p_alpha"], atol=1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_rnnt_decoder_with_align_kd(dev, dtype):
    """kd_type 'align' (rnn_transducer.py:131-135) against the oracle composition on the l4_tiny weights"""
    from oracle import distill as od
    from oracle import model as om
    g = load_kd_golden()
    extra = dict(kd_weight=0.4, kd_type="align", reduce_main_loss_kd=True)
    model, g4 = _model("l4_tiny", extra, dtype, dev, g, "none")
    cfg, sd, _ = load_golden("l4_tiny")
    cfg = SimpleNamespace(**dict(CONFIGS["l4_tiny"], **extra))
    sd = {k: v.clone() for k, v in sd.items()}
    params = {k: v for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    for v in params.values():
        v.requires_grad_(True)
    soft = g["model/soft_rnnt"]
    eouts, elens = om.encoder_forward(sd, cfg, g4["xs"], g4["xlens"], training=True)
    lref, ldref, _, aligns = od.rnnt_decoder_forward_kd_align(sd, cfg, eouts, elens, g4["ys"], g4["ylens"], g4["ys_in"], soft)
    lref.backward()
    model.train()
    loss, ld = model(g4["xs"].to(dev), g4["xlens"], g4["ys"], g4["ylens"], g4["ys_in"], g4["ys_out"],
                     soft_labels=soft.to(dev))
    loss.backward()
    ltol = 1e-3 if dtype == torch.float32 else 5e-2
    assert sorted(ld) == sorted(ldref)
    for k in ldref:
        assert abs(ld[k].item() - float(ldref[k])) < ltol * abs(float(ldref[k])), (k, ld[k].item(), float(ldref[k]))
    grads = {n: p.grad.float().cpu() for n, p in model.named_parameters()}
    for name in ("decoder.output.weight", "decoder.w_dec.weight", "decoder.embed.weight", "encoder.norm.weight",
                 "encoder.transformers.1.feed_forward.w2.weight"):
        a, b = grads[name].flatten(), params[name].grad.flatten()
        cos = torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)
        assert cos > (0.9999 if dtype == torch.float32 else 0.97), (name, cos.item())


def test_aux_ctc_greedy_uses_its_own_head(dev):
    """decode_ctc_weight == 1 short-circuits to greedy CTC on decoder.ctc.output (decoders/transformer.py:176-179)"""
    from oracle import model as om
    cfg, sd, g3 = load_golden("l3_tiny")
    g = load_kd_golden()
    model, _ = _model("l3_tiny", {}, torch.float32, dev, g, "none")
    model.eval()
    hyps, _, _, aligns = model.decode(g3["xs"].to(dev), g3["xlens"], beam_width=1, decode_ctc_weight=1)
    with torch.no_grad():
        eouts, elens = om.encoder_forward(sd, cfg, g3["xs"], g3["xlens"])
        want_h, want_a = om.ctc_greedy(om.linear(sd, "decoder.ctc.output", eouts), elens, cfg.blank_id)
    assert hyps == want_h and aligns == want_a
