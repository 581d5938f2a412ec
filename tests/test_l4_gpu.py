"""RNN-T (config 5) parity on the GPU: transducer lattice kernel vs the path-enumeration-validated
oracle restatement; LSTM / joint / greedy decode / full training step vs goldens produced by the
reference network (l4_tiny)."""
from types import SimpleNamespace

import pytest
import torch

from tests.util import CONFIGS, load_golden, split_ragged

pytestmark = pytest.mark.gpu


def _build(dtype, dev):
    from emoasr_amd.modeling.asr import ASR
    cfg, sd, g = load_golden("l4_tiny")
    model = ASR(SimpleNamespace(**CONFIGS["l4_tiny"]), compute_dtype=dtype)
    model.load_state_dict(sd)
    return model.to(dev), g


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_rnnt_lattice_kernel(dev, dtype):
    from emoasr_amd import ops
    from oracle import rnnt as orn
    torch.manual_seed(0)
    B, T, L, V = 4, 23, 6, 17
    z = (torch.randn(B, T, L + 1, V) * 2).to(dtype)
    labels = torch.randint(1, V, (B, L))
    elens = torch.tensor([23, 17, 9, 1])
    ylens = torch.tensor([6, 3, 0, 2])
    zr = z.float().requires_grad_(True)
    nll_ref = orn.rnnt_nll(torch.log_softmax(zr, -1), labels, elens, ylens, 0)
    nll_ref.mean().backward()
    i32 = lambda t: t.to(torch.int32).to(dev)
    ctx, nll = ops.rnnt_forward(z.to(dev), i32(labels), i32(elens), i32(ylens), 0)
    assert torch.allclose(nll.cpu(), nll_ref.detach(), rtol=1e-4, atol=1e-4), (nll, nll_ref)
    dz = ops.rnnt_grad(z.to(dev), ctx, nll, i32(labels), i32(elens), i32(ylens), 0, 1.0 / B)
    err = (dz.float().cpu() - zr.grad).abs().max().item()
    assert err < (1e-5 if dtype == torch.float32 else 4e-3), err


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_lstm_cell_kernels(dev, dtype):
    from emoasr_amd import ops
    torch.manual_seed(0)
    B, H = 5, 64
    gp = torch.randn(B, 4 * H).to(dtype)
    c0 = torch.randn(B, H)
    gr = gp.float().requires_grad_(True)
    c0r = c0.clone().requires_grad_(True)
    i, f, g, o = gr[:, :H], gr[:, H:2 * H], gr[:, 2 * H:3 * H], gr[:, 3 * H:]
    c1 = torch.sigmoid(f) * c0r + torch.sigmoid(i) * torch.tanh(g)
    h1 = torch.sigmoid(o) * torch.tanh(c1)
    dh, dc_next = torch.randn(B, H).to(dtype), torch.randn(B, H)
    (h1 * dh.float()).sum().backward(retain_graph=True)
    (c1 * dc_next).sum().backward()
    h = torch.empty(B, H, device=dev, dtype=dtype)
    c = torch.empty(B, H, device=dev)
    ga = torch.empty(B, 4 * H, device=dev, dtype=dtype)
    ops.lstm_cell_fwd(gp.to(dev), c0.to(dev), h, c, ga)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert (h.float().cpu() - h1.detach()).abs().max() < tol and (c.cpu() - c1.detach()).abs().max() < tol
    dgp = torch.empty(B, 4 * H, device=dev, dtype=dtype)
    dc = dc_next.clone().to(dev)
    ops.lstm_cell_bwd(dh.to(dev), None, dc, ga, c0.to(dev), c, dgp)
    assert (dgp.float().cpu() - gr.grad).abs().max() < (1e-5 if dtype == torch.float32 else 5e-2)
    assert (dc.cpu() - c0r.grad).abs().max() < (1e-5 if dtype == torch.float32 else 5e-2)


def test_greedy_decode_f32(dev):
    model, g = _build(torch.float32, dev)
    model.eval()
    with torch.no_grad():
        eouts, elens, _ = model.encoder(g["xs"].to(dev), g["xlens"])
        hyps, scores, logits, aligns = model.decoder._greedy(eouts, elens)
        hyps2, s2, l2, a2 = model.decode(g["xs"].to(dev), g["xlens"])
    assert hyps == split_ragged(g["eval/hyps"], g["eval/hyp_lens"])
    assert aligns == split_ragged(g["eval/aligns"], g["eval/align_lens"])
    assert hyps2 == hyps and s2 is None and l2 is None and a2 is None  # decode() discards them (quirk 7)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_greedy_on_the_device_equals_the_launch_chain(dev, dtype):
    """csrc/rnnt_greedy.hip (the whole search of an utterance as ONE cooperative launch: option "rnnt_greedy_coop" 1, the default)
    against the launch chain it replaces (one host round trip per emitted label) and, in f32, the reference's golden hypotheses
    and alignments (the arg-max of EVERY joint evaluation, in order)."""
    from emoasr_amd import lib
    model, g = _build(dtype, dev)
    model.eval()
    eng = model.engine()
    from emoasr_amd import ops
    assert lib.size_query("emoasr_rnnt_greedy_supported", ops._DT[dtype], eng.r_emb, eng.r_H, eng.r_J, model.decoder.output.weight.shape[0],
                          eng.r_nl) == 1

    def run(flag):
        lib.set_option("rnnt_greedy_coop", flag)
        try:
            with torch.no_grad():
                eouts, elens, _ = model.encoder(g["xs"].to(dev), g["xlens"])
                return model.decoder._greedy(eouts, elens)
        finally:
            lib.set_option("rnnt_greedy_coop", 1)

    hyps1, _, _, aligns1 = run(1)
    hyps0, _, _, aligns0 = run(0)
    if dtype == torch.float32:
        assert hyps1 == split_ragged(g["eval/hyps"], g["eval/hyp_lens"])
        assert aligns1 == split_ragged(g["eval/aligns"], g["eval/align_lens"])
        assert hyps1 == hyps0 and aligns1 == aligns0
    else:  # bf16: a near-tie may flip and the sequences diverge from there
        agree = sum(int(a == b) for h1, h0 in zip(hyps1, hyps0) for a, b in zip(h1, h0)) / max(1, sum(len(h) for h in hyps0))
        assert agree > 0.8, (agree, hyps1, hyps0)


@pytest.mark.parametrize("window", [1, 3, 7, 64])
def test_greedy_window_sizes(dev, window):
    """the windowed search (frames scored in batches against an unchanged decoder state) takes the same
    (frame, token) decisions for every window size -- window 1 is the reference's frame-by-frame loop"""
    from emoasr_amd import lib
    model, g = _build(torch.float32, dev)
    model.eval()
    lib.set_option("rnnt_greedy_coop", 0)   # (the windows belong to the launch chain; the device-resident search has none)
    try:
        with torch.no_grad():
            eouts, elens, _ = model.encoder(g["xs"].to(dev), g["xlens"])
            hyps, aligns = model.engine().rnnt_greedy(eouts, elens.tolist(), 0, 2, window=window)
    finally:
        lib.set_option("rnnt_greedy_coop", 1)
    assert hyps == split_ragged(g["eval/hyps"], g["eval/hyp_lens"])
    assert aligns == split_ragged(g["eval/aligns"], g["eval/align_lens"])


def test_beam_search_f32(dev):
    """ALSD beam search (rnn_transducer.py:242-325): the same hypotheses, in the same order, as the
    reference produced for the fitted l4_tiny weights (tests/golden/rnntbeam_tiny.npz)"""
    from tests.util import RNNT_BEAM_WIDTHS, load_rnnt_beam_golden
    model, g = _build(torch.float32, dev)
    model.eval()
    want = load_rnnt_beam_golden()
    with torch.no_grad():
        for b in range(g["xs"].shape[0]):
            n = int(g["xlens"][b])
            for bw in RNNT_BEAM_WIDTHS:
                hyps, scores, logits, aligns = model.decode(g["xs"][b:b + 1, :n].to(dev), g["xlens"][b:b + 1],
                                                            beam_width=bw)
                assert hyps == want[bw][b], (b, bw, hyps, want[bw][b])
                assert scores is None and logits is None and aligns is None


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_train_loss_and_grads(dev, dtype):
    model, g = _build(dtype, dev)
    model.train()
    loss, ld = model(g["xs"].to(dev), g["xlens"], g["ys"], g["ylens"], g["ys_in"], g["ys_out"])
    assert set(ld) == {"loss_rnnt", "loss_ctc", "loss_total"}
    loss.backward()
    ltol = 2e-3 if dtype == torch.float32 else 5e-2
    for k, ref in (("loss_rnnt", "train/loss_rnnt"), ("loss_ctc", "train/loss_ctc"), ("loss_total", "train/loss")):
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
        assert worst < 1e-2, (worst, worst_name)
    else:
        assert cos_min > 0.97, (cos_min, cos_name, worst, worst_name)
