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


# ---------------------------------------------------------------- decode side
def test_log_softmax_topk(dev):
    from emoasr_amd import ops
    torch.manual_seed(0)
    x = torch.randn(5, 1000, device=dev)
    add = torch.randn(5, 1200, device=dev)
    out = ops.log_softmax(x, add=add, mu=0.3)
    ref = torch.log_softmax(x, -1) + 0.3 * add[:, :1000]
    assert torch.allclose(out, ref, atol=1e-5)
    xb = torch.randn(3, 7, 40, device=dev).to(torch.bfloat16)
    out = ops.log_softmax(xb[:, 6])
    assert torch.allclose(out, torch.log_softmax(xb[:, 6].float(), -1), atol=1e-5)
    ref[0, 10] = ref[0, 20] = 99.0  # tie -> lowest index first
    vals, idx, aux = ops.topk(ref, 15, aux=add)
    tv, ti = torch.topk(ref, 15, dim=1)
    assert torch.equal(vals, tv)
    assert idx[0, 0].item() == 10 and idx[0, 1].item() == 20
    assert torch.equal(idx[1:].long(), ti[1:])
    assert torch.equal(aux, torch.gather(add, 1, idx.long()))


def test_ctc_prefix_scorer_kernel(dev):
    import numpy as np
    from emoasr_amd import ops
    from oracle.decoder import CTCPrefixScorer
    rs = np.random.RandomState(0)
    T, V, blank, eos = 37, 40, 0, 2
    x = np.log(rs.dirichlet(np.ones(V) * 0.5, size=T)).astype(np.float32)
    sc = CTCPrefixScorer(x, blank, eos)
    xd = torch.from_numpy(x).to(dev)
    r0 = ops.ctc_prefix_init(xd, blank)
    assert np.allclose(r0.cpu().numpy(), sc.initial_state(), rtol=1e-6, atol=1e-4)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)
    # step 0: one beam [eos]; step 1: two beams extending candidates 1 and 3 of step 0
    c0 = np.array([[5, 2, 0, 7, 9]], dtype=np.int32)
    psi0, st0 = ops.ctc_prefix_score(xd, i32(c0), i32([eos]), i32([0]), blank, eos, init_state=r0)
    ref_psi0, ref_st0 = sc([eos], c0[0], sc.initial_state())
    assert np.allclose(psi0.cpu().numpy()[0], ref_psi0, rtol=1e-5, atol=1e-4)
    c1 = np.array([[5, 2, 7, 11, 0], [7, 7, 2, 5, 30]], dtype=np.int32)
    hyps = [[eos, 5], [eos, 7]]
    psi1, st1 = ops.ctc_prefix_score(xd, i32(c1), i32([5, 7]), i32([1, 1]), blank, eos, prev_states=st0, parent=i32([0, 0]),
                                     pcand=i32([0, 3]))
    for m, (hyp, pc) in enumerate(zip(hyps, [0, 3])):
        ref_psi, ref_st = sc(hyp, c1[m], ref_st0[pc])
        assert np.allclose(psi1.cpu().numpy()[m], ref_psi, rtol=1e-5, atol=1e-4), m
        got = st1.cpu().numpy()[m][:, 1:]
        assert np.allclose(got, ref_st[:, 1:], rtol=1e-5, atol=1e-4), m


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_lm_predict(dev, dtype):
    from emoasr_amd.modeling.lm import LM
    cfg, sd, g = load_golden("l3_tiny")
    lm = LM(SimpleNamespace(**LM_CFG), compute_dtype=dtype)
    lm.load_state_dict(lm_state(g))
    lm = lm.to(dev).eval()
    lp, states = lm.predict(g["lm_test/ys"], g["lm_test/ylens"])
    assert states is None and lp.shape == g["lm_test/logp"].shape
    err = (lp.cpu() - g["lm_test/logp"]).abs().max().item()
    assert err < (2e-3 if dtype == torch.float32 else 0.15), err


def test_joint_beam_search_f32(dev):
    """hypotheses identical to the reference's, scores within 1e-3, for attention-only, +CTC, +CTC+LM, +LM"""
    from emoasr_amd.modeling.lm import LM
    model, g = _build(torch.float32, dev)
    model.eval()
    lm = LM(SimpleNamespace(**LM_CFG), compute_dtype=torch.float32)
    lm.load_state_dict(lm_state(g))
    lm = lm.to(dev).eval()
    for si, st in enumerate(DECODE_SETTINGS):
        for b in range(2):
            n = int(g["xlens"][b])
            hyps, scores, _, _ = model.decode(g["xs"][b:b + 1, :n].to(dev), g["xlens"][b:b + 1], lm=lm, **st)
            want = split_ragged(g[f"decode/{si}/{b}/hyps"], g[f"decode/{si}/{b}/lens"])
            assert hyps == want, (si, b, hyps, want)
            ref = g[f"decode/{si}/{b}/scores"].numpy()
            assert max(abs(a - c) for a, c in zip(scores, ref)) < 1e-2 + 1e-3 * abs(ref).max(), (si, b, scores, ref)


def test_joint_beam_search_bf16_runs(dev):
    from emoasr_amd.modeling.lm import LM
    model, g = _build(torch.bfloat16, dev)
    model.eval()
    lm = LM(SimpleNamespace(**LM_CFG), compute_dtype=torch.bfloat16)
    lm.load_state_dict(lm_state(g))
    lm = lm.to(dev).eval()
    n = int(g["xlens"][0])
    hyps, scores, _, _ = model.decode(g["xs"][:1, :n].to(dev), g["xlens"][:1], lm=lm, **DECODE_SETTINGS[2])
    ref = g["decode/2/0/scores"].numpy()
    assert len(hyps) >= 1 and abs(scores[0] - ref[0]) < 0.05 * abs(ref[0])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_device_beam_search_equals_host_bookkeeping(dev, dtype, monkeypatch):
    """the device-resident search (K / V caches, one position per step, emoasr_beam_update) against the host bookkeeping
    over whole-prefix recomputation: same hypotheses in the same order, scores to the last bits of the f32 candidate
    arithmetic, for all four score combinations, with and without a length bonus; and the device path is the one taken"""
    from emoasr_amd.modeling import beam_search_device as bsd
    from emoasr_amd.modeling.lm import LM
    model, g = _build(dtype, dev)
    model.eval()
    lm = LM(SimpleNamespace(**LM_CFG), compute_dtype=dtype)
    lm.load_state_dict(lm_state(g))
    lm = lm.to(dev).eval()
    calls = []
    orig = bsd.joint_beam_search_device
    monkeypatch.setattr(bsd, "joint_beam_search_device", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    for si, st in enumerate(DECODE_SETTINGS):
        for lw in (0.0, 0.3):
            kw = dict(st, len_weight=lw)
            for b in range(2):
                n = int(g["xlens"][b])
                x, xl = g["xs"][b:b + 1, :n].to(dev), g["xlens"][b:b + 1]
                monkeypatch.setenv("EMOASR_DEVICE_BEAM", "1")
                monkeypatch.setenv("EMOASR_BEAM_GRAPH", "1" if b == 0 else "0")   # the step as HIP graphs / as plain launches
                k0 = len(calls)
                hyps, scores, _, _ = model.decode(x, xl, lm=lm, **kw)
                assert len(calls) == k0 + 1, "device path not taken"
                monkeypatch.setenv("EMOASR_DEVICE_BEAM", "0")
                hyps_h, scores_h, _, _ = model.decode(x, xl, lm=lm, **kw)
                assert len(calls) == k0 + 1
                if dtype == torch.float32:
                    assert hyps == hyps_h, (si, lw, b, hyps, hyps_h)
                    assert max(abs(a - c) for a, c in zip(scores, scores_h)) < 1e-4, (scores, scores_h)
                else:  # bf16: cached K / V and recomputed prefixes round alike, but near-ties may still flip
                    assert hyps[0] == hyps_h[0] and abs(scores[0] - scores_h[0]) < 2e-2 * abs(scores_h[0]) + 1e-3


@pytest.mark.parametrize("option", ["decode_coop", "decode_coop_merge3", "decode_coop_merge0"])
def test_alternative_decode_step_kernels_agree(dev, option):
    """the forms of the cached decode steps -- csrc/decode_coop.hip (the default: one launch of 16 cooperating workgroups per
    network, grid barriers between the stages), and the two measured-and-not-kept ones, csrc/decode_wg.hip (one workgroup per
    network) and csrc/rowlin.hip (LayerNorm / cache append folded into small-M kernels) -- give the same best hypothesis as the
    plain launch chain (bf16: scores to 2 %)"""
    from emoasr_amd import lib
    from emoasr_amd.modeling.lm import LM
    model, g = _build(torch.bfloat16, dev)
    model.eval()
    lm = LM(SimpleNamespace(**LM_CFG), compute_dtype=torch.bfloat16)
    lm.load_state_dict(lm_state(g))
    lm = lm.to(dev).eval()
    try:
        for u in range(min(3, len(g["xlens"]))):
            n = int(g["xlens"][u])
            x, xl = g["xs"][u:u + 1, :n].to(dev), g["xlens"][u:u + 1]
            lib.set_option("decode_coop", 0)
            ref_h, ref_s, _, _ = model.decode(x, xl, lm=lm, **DECODE_SETTINGS[2])
            if option.startswith("decode_coop_merge"):   # projection + self-attention as one stage in both / in neither stack
                lib.set_option("decode_coop", 1)
                lib.set_option("decode_coop_merge", int(option[-1]))
            else:
                lib.set_option(option, 1)
            hyps, scores, _, _ = model.decode(x, xl, lm=lm, **DECODE_SETTINGS[2])
            if not option.startswith("decode_coop"):
                lib.set_option(option, 0)
            assert lib.size_query("emoasr_decode_coop_status") == 0
            assert hyps[0] == ref_h[0] and abs(scores[0] - ref_s[0]) < 2e-2 * abs(ref_s[0]) + 1e-3, (u, hyps[0], ref_h[0], scores[0], ref_s[0])
    finally:
        if not option.startswith("decode_coop"):
            lib.set_option(option, 0)
        lib.set_option("decode_coop", 1)
        lib.set_option("decode_coop_merge", 1)
