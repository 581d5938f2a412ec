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


@pytest.mark.parametrize("graph", ["1", "0"], ids=["graph", "chain"])
def test_beam_search_f32(dev, graph, monkeypatch):
    """ALSD beam search (rnn_transducer.py:242-325): the same hypotheses, in the same order, as the
    reference produced for the fitted l4_tiny weights (tests/golden/rnntbeam_tiny.npz) -- with the expansion round replayed
    from a HIP graph (the default) and as the launch chain"""
    from tests.util import RNNT_BEAM_WIDTHS, load_rnnt_beam_golden
    monkeypatch.setenv("EMOASR_RNNT_BEAM_GRAPH", graph)
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


@pytest.mark.parametrize("mfma", [1, 0], ids=["mfma", "valu"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("nb", [1, 5, 16])
def test_beam_round_kernels(dev, dtype, nb, mfma):
    """csrc/rnnt_beam.hip against a plain PyTorch fp32 expression of the same round (rnn_transducer.py:242-325: LSTM step of every
    live hypothesis from / to slot-addressed states, joint input, log-softmax + blank + top-k record): bf16 on the matrix cores
    (`mfma`) and on the VALU kernels that also serve f32"""
    from ctypes import c_void_p
    from emoasr_amd import lib, ops
    if dtype == torch.float32 and mfma:
        pytest.skip("the matrix-core form is bf16")
    lib.set_option("rnnt_beam_mfma", mfma)
    try:
        torch.manual_seed(nb)
        E, H, J, V, POOL, Tm, k = 64, 96, 48, 200, 40, 7, 4
        r = lambda *s, sc=1.0: (torch.randn(*s, device=dev) * sc).to(dtype)
        emb, w_ih, w_hh = r(V, E), r(4 * H, E, sc=E ** -0.5), r(4 * H, H, sc=H ** -0.5)
        bias = torch.randn(4 * H, device=dev) * 0.1
        ph, pc = r(POOL, H, sc=0.5), torch.randn(POOL, H, device=dev) * 0.5
        w_dec, b_dec, e_all = r(J, H, sc=H ** -0.5), torch.randn(J, device=dev) * 0.1, r(Tm, J)
        ids = torch.randint(0, V, (nb,), device=dev)
        src = torch.randperm(16, device=dev)[:nb]
        dst = 20 + torch.arange(nb, device=dev)
        t = torch.tensor([3], device=dev)
        # reference (f32 math on the values as stored)
        x, h0, c0 = emb.float()[ids], ph.float()[src], pc[src]
        g = x @ w_ih.float().t() + bias + h0 @ w_hh.float().t()
        i_, f_, g_, o_ = g.chunk(4, 1)
        c1 = torch.sigmoid(f_) * c0 + torch.sigmoid(i_) * torch.tanh(g_)
        h1 = torch.sigmoid(o_) * torch.tanh(c1)
        ph_k, pc_k = ph.clone(), pc.clone()
        ptr = lambda tt: c_void_p(tt.data_ptr())
        lib.call("emoasr_rnnt_beam_lstm", ops.dt(emb), nb, E, H, ptr(emb), E, ptr(ids), ptr(w_ih), ptr(w_hh), ptr(bias), ptr(ph_k),
                 ptr(pc_k), ptr(src), ptr(dst), None, None, 0, ops._stream())
        tol = 1e-5 if dtype == torch.float32 else 3e-2
        assert (ph_k[dst].float() - h1).abs().max().item() < tol and (pc_k[dst] - c1).abs().max().item() < tol
        keep = torch.ones(POOL, dtype=torch.bool, device=dev)
        keep[dst] = False
        assert torch.equal(ph_k[keep], ph[keep]) and torch.equal(pc_k[keep], pc[keep])   # nothing but the destination slots moved
        hj = torch.zeros(16, J, device=dev, dtype=dtype)
        lib.call("emoasr_rnnt_beam_joint", ops.dt(emb), nb, H, J, Tm, ptr(ph_k), ptr(dst), ptr(w_dec), ptr(b_dec), ptr(e_all), ptr(t),
                 ptr(hj), ops._stream())
        hj_ref = torch.tanh(e_all.float()[3] + ph_k[dst].float() @ w_dec.float().t() + b_dec)
        assert (hj[:nb].float() - hj_ref).abs().max().item() < tol
        for Vp in (V, 1000, 1500):   # <= 1024: the register-resident row; above: the LDS row
            logits = r(nb, Vp, sc=2.0)
            out = torch.zeros(16, 1 + 2 * 16, device=dev)
            lib.call("emoasr_rnnt_beam_pick", ops.dt(logits), nb, Vp, k, 0, ptr(logits), Vp, ptr(out), out.stride(0), ops._stream())
            lp = torch.log_softmax(logits.float(), -1)
            vals, idx = torch.topk(lp[:, 1:], k, dim=-1)
            assert torch.allclose(out[:nb, 0], lp[:, 0], atol=1e-5), Vp
            assert torch.allclose(out[:nb, 1:1 + k], vals, atol=1e-5), Vp
            got = out[:nb, 1 + k:1 + 2 * k].long()
            assert torch.equal(lp[:, 1:].gather(1, got), vals), Vp   # (bf16 logits tie: torch.topk's order among equals is unspecified)
            tie = vals[:, 1:] == vals[:, :-1]
            assert bool((got[:, 1:][tie] > got[:, :-1][tie]).all()), Vp   # ties -> lowest index first
    finally:
        lib.set_option("rnnt_beam_mfma", 1)


def test_beam_search_bf16_fused_round_agrees_with_the_chain(dev, monkeypatch):
    """bf16 (what bench.py's beam4_rtf runs): the graph form with the fused round kernels returns the launch chain's hypotheses
    on the fitted golden model (near-ties may flip a late label: token agreement > 0.9)"""
    model, g = _build(torch.bfloat16, dev)
    model.eval()
    outs = {}
    with torch.no_grad():
        for form in ("1", "0"):
            monkeypatch.setenv("EMOASR_RNNT_BEAM_GRAPH", form)
            res = []
            for b in range(g["xs"].shape[0]):
                n = int(g["xlens"][b])
                res.append(model.decode(g["xs"][b:b + 1, :n].to(dev), g["xlens"][b:b + 1], beam_width=4)[0][0])
            outs[form] = res
    same = sum(int(a == c) for ha, hc in zip(outs["1"], outs["0"]) for a, c in zip(ha, hc))
    tot = sum(max(len(ha), len(hc)) for ha, hc in zip(outs["1"], outs["0"]))
    assert same / max(tot, 1) > 0.9, (outs["1"], outs["0"])


def test_beam_search_graph_scores_on_fresh_engine(dev):
    """the graph form's warm-up before each capture must not touch the live LSTM state pool (it once ran the round's body with
    the PREVIOUS round's control words: slot 0, the zero state, was overwritten): on a FRESH engine, first utterance first and
    beam widths ascending -- every search captures new graphs -- the per-hypothesis scores equal the launch chain's"""
    model, g = _build(torch.float32, dev)
    model.eval()
    chain_model, _ = _build(torch.float32, dev)
    chain_model.eval()
    with torch.no_grad():
        for b in range(g["xs"].shape[0]):
            n = int(g["xlens"][b])
            for bw in (1, 2, 3, 4, 5):
                outs = []
                for m, form in ((model, "_rnnt_beam_search_graph"), (chain_model, "_rnnt_beam_search_chain")):
                    eouts, elens, _ = m.encoder(g["xs"][b:b + 1, :n].to(dev), g["xlens"][b:b + 1])
                    eng = m.engine()
                    outs.append(getattr(eng, form)(eouts[:, :int(elens[0])], bw, m.decoder.blank_id, m.decoder.eos_id,
                                                   return_scores=True))
                (h_g, s_g), (h_c, s_c) = outs
                assert h_g == h_c, (b, bw, h_g, h_c)
                assert max(abs(a - c) for a, c in zip(s_g, s_c)) < 1e-4, (b, bw, s_g, s_c)


def test_beam_search_outside_the_fused_round_limits(dev):
    """a vocabulary wider than the fused round's pick kernel holds in LDS (V * 4 B > 60 KB, csrc/rnnt_beam.hip): the graph form must
    decode through the launch chain's body (as before round 5) instead of failing in its warm-up -- same hypotheses and scores as
    _rnnt_beam_search_chain"""
    from emoasr_amd.modeling.asr import ASR
    cfg = dict(CONFIGS["l4_tiny"], vocab_size=16001)
    torch.manual_seed(5)
    models = []
    for _ in range(2):
        m = ASR(SimpleNamespace(**cfg), compute_dtype=torch.float32)
        if models:
            m.load_state_dict(models[0].state_dict())
        models.append(m)
    models = [m.to(dev).eval() for m in models]
    _, _, g = load_golden("l4_tiny")
    with torch.no_grad():
        n = int(g["xlens"][0])
        outs = []
        for m, form in zip(models, ("_rnnt_beam_search_graph", "_rnnt_beam_search_chain")):
            eouts, elens, _ = m.encoder(g["xs"][:1, :n].to(dev), g["xlens"][:1])
            outs.append(getattr(m.engine(), form)(eouts[:, :int(elens[0])], 3, m.decoder.blank_id, m.decoder.eos_id,
                                                  return_scores=True))
        (h_g, s_g), (h_c, s_c) = outs
        assert h_g == h_c, (h_g, h_c)
        assert max(abs(a - c) for a, c in zip(s_g, s_c)) < 1e-4, (s_g, s_c)


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


def test_fused_output_layer_against_oracle_and_materialised_path(dev):
    """The transducer's output layer without the [B,T,U,V] logits (ops.rnnt_head_forward / rnnt_coef / rnnt_head_grad: soft-max
    partials and the blank / label gathers in the GEMM's epilogue, the gradient rows recomputed chunk by chunk) at the L4 widths
    (J = 512, V = 1000), ragged lengths, an utterance without labels and one frame-starved (infeasible) utterance:
    (a) kernel level against the ORACLE (oracle/rnnt.py on f32 logits = h W^T + b): nll 2e-3, d(loss)/d(logits) rows 3e-3 of
        the largest entry -- the fused path reduces f32 accumulators, it never rounds the logits to bf16;
    (b) engine level against the materialised path on the same model: losses 2e-3, every decoder / encoder gradient cosine
        0.999, the gradient walked in six chunks + a remainder; the training forward returns logits = None and its peak
        memory stays below the materialised path's by most of the logits tensor's size (what replaces it: the 2048-row gradient
        chunk, the soft-max partials and the row constants).
    Reference: asr/modeling/decoders/rnn_transducer.py:101-115,147-156."""
    from emoasr_amd import ops
    from emoasr_amd.modeling.asr import ASR
    from oracle import rnnt as orn
    torch.manual_seed(0)
    # ---- (a) kernels
    B, T, L, V, J = 4, 37, 9, 1000, 512
    U = L + 1
    h = torch.tanh(torch.randn(B * T * U, J)).to(torch.bfloat16)
    w = (torch.randn(V, J) / J ** 0.5 * 3).to(torch.bfloat16)
    bias = torch.randn(V) * 0.5
    bias[0] += 2.0
    labels = torch.randint(1, V, (B, L))
    elens, ylens = torch.tensor([37, 30, 11, 2]), torch.tensor([9, 5, 0, 7])   # the last one: 2 frames for 7 labels is FEASIBLE in a
    # transducer (labels are emitted without consuming frames); an empty utterance (elens 0) is covered by test_rnnt_lattice_kernel
    z = (h.float() @ w.float().t() + bias).view(B, T, U, V).requires_grad_(True)
    nll_ref = orn.rnnt_nll(torch.log_softmax(z, -1), labels, elens, ylens, 0)
    nll_ref.mean().backward()
    i32 = lambda t: t.to(torch.int32).to(dev)
    hd, wd, bd = h.to(dev), w.to(dev), bias.to(dev)
    with ops.stream_scope():
        ctx, nll = ops.rnnt_head_forward(hd, wd, bd, B, T, U, i32(labels), i32(elens), i32(ylens), 0)
        coef, ycol = ops.rnnt_coef(ctx, nll, i32(labels), i32(elens), i32(ylens), 1.0 / B)
        dz = torch.full((B * T * U, V), float("nan"), device=dev, dtype=torch.bfloat16)
        for r0 in range(0, B * T * U, 500):   # uneven chunks
            n = min(500, B * T * U - r0)
            ops.rnnt_head_grad(hd[r0:r0 + n], wd, bd, coef[r0:r0 + n], ycol[r0:r0 + n], 0, dz[r0:r0 + n])
    assert torch.allclose(nll.cpu(), nll_ref.detach(), rtol=2e-3, atol=2e-3), (nll, nll_ref)
    err = (dz.float().cpu().view(B, T, U, V) - z.grad).abs().max().item() / z.grad.abs().max().item()
    print(f"[measured] fused transducer head: nll {nll.tolist()} vs {nll_ref.tolist()}, gradient rows max err {err:.2e} of max")
    assert err < 3e-3, err
    # ---- (b) engine: fused against materialised
    cfg = dict(input_layer="conv2d", feat_dim=80, num_framestacks=1, encoder_type="conformer", pos_encode_type="rel",
               enc_hidden_size=256, enc_num_attention_heads=4, enc_num_layers=2, enc_intermediate_size=512,
               dropout_enc_rate=0.0, dropout_attn_rate=0.0, vocab_size=1000, blank_id=0, eos_id=2, kd_weight=0,
               decoder_type="rnn_transducer", embedding_size=256, dec_hidden_size=512, dec_num_layers=2, joint_hidden_size=512,
               dropout_emb_rate=0.0, dropout_dec_rate=0.0, mtl_ctc_weight=0.3, lsm_prob=0.0)
    model = ASR(SimpleNamespace(**cfg), compute_dtype=torch.bfloat16).to(dev).train()
    eng = model.engine()
    g = torch.Generator().manual_seed(5)
    xlens, yl = torch.tensor([403, 363, 303, 250, 203, 99]), torch.tensor([20, 12, 10, 9, 1, 4])
    xs = torch.randn(6, 403, 80, generator=g)
    ys = torch.randint(3, 1000, (6, 20), generator=g)
    for b in range(6):
        xs[b, xlens[b]:] = 0
        ys[b, yl[b]:] = 2
    eos = torch.full((6, 1), 2)
    ys_in, ys_out = torch.cat([eos, ys], 1), torch.cat([ys, eos], 1)
    res = {}
    for fused in (False, True):
        eng.rnnt_fused, eng.rnnt_chunk = fused, 2048    # cells: 6 x 100 x 21 = 12 600 -> six chunks + a remainder
        eng.arena.grad.zero_()
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        loss, ld = model(xs.to(dev), xlens, ys, yl, ys_in, ys_out)
        loss.backward()
        torch.cuda.synchronize()
        res[fused] = ({k: float(v) for k, v in ld.items()}, eng.arena.grad.clone(), torch.cuda.max_memory_allocated() - base)
    (l0, g0, m0), (l1, g1, m1) = res[False], res[True]
    for k in l0:
        assert abs(l0[k] - l1[k]) < 2e-3 * abs(l0[k]) + 1e-4, (k, l0, l1)
    A = eng.arena
    gmax = g0.abs().max().item()
    for name in A.names:
        o, k = A.offsets[name], A.pviews[name].numel()
        a, b_ = g1[o:o + k], g0[o:o + k]
        if b_.abs().max() < 1e-3 * gmax:
            assert a.abs().max() < 4e-3 * gmax, name
            continue
        cos = (torch.dot(a, b_) / (a.norm() * b_.norm() + 1e-30)).item()
        assert cos > 0.999, (name, cos)
    logits_bytes = 6 * 100 * 21 * 1000 * 2
    print(f"[measured] peak memory of one training step: materialised {m0 / 1e6:.1f} MB, fused {m1 / 1e6:.1f} MB "
          f"(the logits tensor: {logits_bytes / 1e6:.1f} MB)")
    assert m1 < m0 - 0.7 * logits_bytes, (m0, m1, logits_bytes)
    eng.rnnt_fused = True
    out = model.decoder(*model.encoder(xs.to(dev), xlens)[:2], None, ys, yl, ys_in, ys_out)
    assert out[2] is None                      # training forward: no 4-D logits
    model.decoder.return_logits = True
    out = model.decoder(*model.encoder(xs.to(dev), xlens)[:2], None, ys, yl, ys_in, ys_out)
    assert out[2] is not None and tuple(out[2].shape) == (6, 100, 21, 1000)


def test_prediction_network_of_all_micro_batches_in_one_pass(dev):
    """train.train_group on a transducer stacks the ENCODER and (round 4) the PREDICTION NETWORK of the micro-batches: the LSTM
    recurrences of all of them in one grouped cooperative launch per layer (csrc/lstm_coop.hip: groups of 64 sequences), labels
    padded to the longest.  Against the same group with the prediction network run per micro-batch (decoder.prediction_stacked
    switched off): losses 1e-3, every gradient cosine 0.999 / norm 1 % -- the recurrences are bit-identical per sequence, the
    weight-gradient products sum over differently grouped rows.  70 + 66 + 3 sequences: three groups, ragged last one."""
    from emoasr_amd.modeling.asr import ASR
    from emoasr_amd.train import ArenaAdam, train_group
    cfg = dict(input_layer="conv2d", feat_dim=80, num_framestacks=1, encoder_type="conformer", pos_encode_type="rel",
               enc_hidden_size=256, enc_num_attention_heads=4, enc_num_layers=2, enc_intermediate_size=512,
               dropout_enc_rate=0.0, dropout_attn_rate=0.0, vocab_size=1000, blank_id=0, eos_id=2, kd_weight=0,
               decoder_type="rnn_transducer", embedding_size=256, dec_hidden_size=512, dec_num_layers=2, joint_hidden_size=512,
               dropout_emb_rate=0.0, dropout_dec_rate=0.0, mtl_ctc_weight=0.3, lsm_prob=0.0, accum_grad=3, clip_grad_norm=5.0)
    params = SimpleNamespace(**cfg)
    g = torch.Generator().manual_seed(11)
    datas = []
    for B, T, L in ((70, 83, 7), (66, 99, 12), (3, 140, 5)):
        xlens = torch.randint(T // 2, T + 1, (B,), generator=g)
        xlens[0] = T
        ylens = torch.randint(1, L + 1, (B,), generator=g)
        ylens[1 % B] = L
        xs = torch.randn(B, T, 80, generator=g)
        ys = torch.randint(3, 1000, (B, L), generator=g)
        for b in range(B):
            xs[b, xlens[b]:] = 0
            ys[b, ylens[b]:] = 2
        eos = torch.full((B, 1), 2)
        datas.append(dict(xs=xs, xlens=xlens, ys=ys, ylens=ylens, ys_in=torch.cat([eos, ys], 1), ys_out=torch.cat([ys, eos], 1)))
    res = {}
    for stacked_pred in (True, False):
        torch.manual_seed(3)
        model = ASR(params, compute_dtype=torch.bfloat16).to(dev).train()
        eng = model.engine()
        if not stacked_pred:
            model.decoder.prediction_stacked = lambda *a, **k: None
        else:
            calls = []
            orig = eng.rnnt_recurrency
            eng.rnnt_recurrency = lambda ids, *a, **k: (calls.append(tuple(ids.shape)), orig(ids, *a, **k))[1]
        opt = ArenaAdam(eng.arena, lambda s: 0.0, clip_grad_norm=5.0)   # lr 0: the step leaves the weights, the gradients are kept below
        grads = {}
        step = opt.step
        opt.step = lambda: (grads.update(g=eng.arena.grad.clone()), step())[1]
        dicts = train_group(model, opt, datas, params, dev)
        res[stacked_pred] = ([d["loss_total"] for d in dicts], grads["g"], eng.arena)
        if stacked_pred:
            assert calls == [(13, 139)], calls   # ONE prediction-network pass: longest label row + <sos>, all sequences
    (l1, g1, A), (l0, g0, _) = res[True], res[False]
    for a, b in zip(l1, l0):
        assert abs(a - b) < 1e-3 * abs(b), (l1, l0)
    gmax = g0.abs().max().item()
    for name in A.names:
        o, k = A.offsets[name], A.pviews[name].numel()
        a, b = g1[o:o + k], g0[o:o + k]
        if b.abs().max() < 1e-3 * gmax:
            assert a.abs().max() < 2e-3 * gmax, name
            continue
        cos = torch.nn.functional.cosine_similarity(a, b, dim=0).item()
        assert cos > 0.999, (name, cos)
        assert abs(a.norm().item() / b.norm().item() - 1) < 1e-2, (name, a.norm().item(), b.norm().item())
