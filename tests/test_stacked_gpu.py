"""Stacked micro-batches (engine.ctc_train_stacked / train.train_group): the `accum_grad` micro-batches of one optimizer step
through the engine in ONE pass -- rows concatenated for the row-wise kernels, attention / convolution padding / BatchNorm
statistics per micro-batch -- against the same micro-batches run one after the other through the module API, which is what
the reference does (asr/train_asr.py:106-128) and what the goldens pin.

bf16, dropout 0: losses equal to 1e-3, every parameter gradient to cosine 0.999 and 2 % of its norm (summation order differs:
split-K slices and LayerNorm partial sums follow the row count), BatchNorm running statistics to 1e-4, num_batches_tracked
exactly; then three optimizer steps of train.train vs the one-by-one loop."""
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = dict(input_layer="conv2d", feat_dim=80, num_framestacks=1, encoder_type="conformer", decoder_type="ctc",
           pos_encode_type="rel", enc_hidden_size=256, enc_num_attention_heads=4, enc_num_layers=3,
           enc_intermediate_size=512, dropout_enc_rate=0.0, dropout_attn_rate=0.0, vocab_size=96, blank_id=0,
           eos_id=2, kd_weight=0, lr_schedule_type="noam", learning_rate=1.0, num_warmup_steps=10, accum_grad=3,
           clip_grad_norm=5.0, weight_decay=1e-6, log_step=100)


def _batch(seed, xlens):
    g = torch.Generator().manual_seed(seed)
    xlens = torch.tensor(xlens)
    ylens = torch.clamp(xlens // 40, min=1)
    B, T, L = len(xlens), int(xlens.max()), int(ylens.max())
    xs = torch.randn(B, T, 80, generator=g)
    ys = torch.randint(3, CFG["vocab_size"], (B, L), generator=g)
    for b in range(B):
        xs[b, xlens[b]:] = 0
        ys[b, ylens[b]:] = 2
    return dict(xs=xs, xlens=xlens, ys=ys, ylens=ylens, ys_in=None, ys_out=None)


def _micro_batches(shape="regular"):
    # different batch sizes AND different padded lengths (T' = 49 / 99 / 74), ragged utterances inside each
    if shape == "extreme":   # ... down to segments of one encoder frame (7 input frames) next to a 50-frame one
        return [_batch(1, [203, 35]), _batch(2, [7, 7]), _batch(3, [11, 7, 7])]
    return [_batch(1, [203, 187, 150, 96]), _batch(2, [403, 380]), _batch(3, [303, 290, 221])]


def _model(dev, mode=torch.bfloat16, **over):
    from emoasr_amd.modeling.asr import ASR
    torch.manual_seed(0)
    model = ASR(SimpleNamespace(**dict(CFG, **over)), compute_dtype=mode)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "batch_norm" in n or ".norm" in n:
                p.add_(0.05 * torch.randn_like(p))
    return model.to(dev).train()


def _cos(a, b):
    a, b = a.flatten().float(), b.flatten().float()
    return (torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)).item()


@pytest.mark.parametrize("shape", ["regular", "extreme"])
@pytest.mark.parametrize("mode", [torch.bfloat16, torch.float32, "f32x3"], ids=["bf16", "f32", "f32x3"])
def test_stacked_pass_equals_the_separate_passes(dev, mode, shape):
    """f32 / f32x3 (the layer backward in C++ with the materialised attention backward per micro-batch): the same comparison at
    losses 1e-5, gradient cosine 0.99999 and 1e-3 of the norm, running statistics 1e-5"""
    model = _model(dev, mode)
    f32 = mode is not torch.bfloat16
    ltol, cbar, ntol, rtol = (1e-5, 0.99999, 1e-3, 1e-5) if f32 else (1e-3, 0.999, 2e-2, 2e-3)
    eng = model.engine()
    assert eng.stacked_ok()
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    datas = _micro_batches(shape)
    n = len(datas)
    # ---- one after the other (module API, autograd): loss / n each
    eng.arena.grad.zero_()
    want_losses = []
    for data in datas:
        loss, _ = model(data["xs"].to(dev), data["xlens"], data["ys"], data["ylens"], None, None)
        (loss / n).backward()
        want_losses.append(loss.item())
    want_grad = eng.arena.grad.clone()
    want_bufs = {k: v.detach().clone() for k, v in model.state_dict().items() if "running" in k or "tracked" in k}
    # ---- stacked, from the same state
    model.load_state_dict(sd0)
    eng.arena.grad.zero_()
    batches = [(d["xs"].to(dev), [int(v) for v in d["xlens"]], d["ys"], [int(v) for v in d["ylens"]]) for d in datas]
    losses = eng.ctc_train_stacked(batches, 0)
    torch.cuda.synchronize()
    got = losses.tolist()
    for a, b in zip(got, want_losses):
        assert abs(a - b) < ltol * abs(b), (got, want_losses)
    A = eng.arena
    worst = (1.0, "")
    gmax = want_grad.abs().max().item()
    for name in A.names:
        o, k = A.offsets[name], A.pviews[name].numel()
        g, w = A.grad[o:o + k], want_grad[o:o + k]
        if w.abs().max() == 0:
            assert g.abs().max() == 0, name
            continue
        if w.abs().max() < 1e-3 * gmax:
            # a gradient that is zero in exact arithmetic (linear_k.bias: the soft-max does not see a constant added to every key's
            # score) is rounding noise in both passes: its direction means nothing, its size must stay noise
            assert g.abs().max() < 2e-3 * gmax, (name, g.abs().max().item(), gmax)
            continue
        cos = _cos(g, w)
        worst = min(worst, (cos, name))
        assert cos > cbar, (name, cos)
        assert abs(g.norm().item() / w.norm().item() - 1) < ntol, (name, g.norm().item(), w.norm().item())
    print("stacked vs separate: worst gradient cosine", worst)
    for k, v in model.state_dict().items():
        if "tracked" in k:
            assert torch.equal(v, want_bufs[k]), k                 # one BatchNorm update per micro-batch
        elif "running" in k:
            err = (v - want_bufs[k]).abs().max().item() / (want_bufs[k].abs().max().item() + 1e-12)
            # (bf16 activations; the separate passes' small attention launches split the keys over four waves, the stacked launch
            # does not: the two differ by the order of the soft-max sums; a few bf16 roundings that flip move the mean of a
            # channel by this much: 1.2e-4 .. 5.8e-4 measured over library builds)
            assert err < rtol, (k, err)


@pytest.mark.parametrize("mode", [torch.bfloat16, "f32x3"], ids=["bf16", "f32x3"])
def test_stacked_pass_with_dropout_is_reproducible_and_finite(dev, mode):
    model = _model(dev, mode, dropout_enc_rate=0.1, dropout_attn_rate=0.1)
    eng = model.engine()
    datas = _micro_batches()
    same = [datas[0], datas[0]]   # the same micro-batch twice: independent dropout masks -> different losses
    batches = [(d["xs"].to(dev), [int(v) for v in d["xlens"]], d["ys"], [int(v) for v in d["ylens"]]) for d in same]
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}

    def run():
        model.load_state_dict(sd0)
        eng.step_count = 11
        eng.arena.grad.zero_()
        losses = eng.ctc_train_stacked(batches, 0)
        return losses.tolist(), eng.arena.grad.clone()

    l1, g1 = run()
    l2, g2 = run()
    assert l1 == l2 and l1[0] != l1[1], (l1, l2)
    assert torch.isfinite(g1).all() and g1.abs().max() > 0
    assert (g1 - g2).abs().max().item() < 2e-3 * g1.abs().max().item()   # (float atomics in the weight gradients)


@pytest.mark.parametrize("mode", [torch.bfloat16, "f32x3"], ids=["bf16", "f32x3"])
def test_weight_gradients_on_the_side_stream_equal_the_inline_launches(dev, mode):
    """option wgrad_side: every layer's grouped weight-gradient launch on the side stream, two workspaces in turn -- the same
    gradients as the inline launches (float atomics aside), gradient hooks one layer behind the sweep and still in order"""
    from emoasr_amd import lib
    model = _model(dev, mode, dropout_enc_rate=0.1, dropout_attn_rate=0.1)
    eng = model.engine()
    datas = _micro_batches()
    batches = [(d["xs"].to(dev), [int(v) for v in d["xlens"]], d["ys"], [int(v) for v in d["ylens"]]) for d in datas]
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    res = {}
    try:
        for side in (False, True, False):
            model.load_state_dict(sd0)
            eng.step_count = 5
            eng.arena.grad.zero_()
            if eng._layer_rt is not None:
                eng._layer_rt.wgrad_side = side
            lib.set_option("wgrad_side", int(side))
            seen = []
            eng.grad_hook = seen.append
            losses = eng.ctc_train_stacked(batches, 0)
            if eng._layer_rt.wgrad_side != side:   # (the runtime was created by this first pass: switch and repeat)
                eng._layer_rt.wgrad_side = side
            torch.cuda.synchronize()
            res.setdefault(side, []).append((losses.tolist(), eng.arena.grad.clone(), seen))
    finally:
        eng.grad_hook = None
        lib.set_option("wgrad_side", 0)
        if eng._layer_rt is not None:
            eng._layer_rt.wgrad_side = False
    (l0, g0, h0), (l1, g1, h1) = res[False][0], res[True][0]
    assert l0 == l1
    gmax = g0.abs().max().item()
    noise = (res[False][1][1] - g0).abs().max().item()     # two inline runs: the float-atomic reordering alone
    assert (g1 - g0).abs().max().item() <= max(4 * noise, 1e-6 * gmax), ((g1 - g0).abs().max().item(), noise, gmax)
    # hooks: descending offsets; the side-stream sweep reports every layer one call later and layer 0 after the join
    assert h0 == sorted(h0, reverse=True) and h1 == sorted(h1, reverse=True)
    assert set(h1) <= set(h0) and min(h1) == min(h0)


def test_train_loop_takes_the_stacked_path(dev, monkeypatch):
    """train.train with accum_grad 3: two optimizer steps through train_group == six train_step calls"""
    from emoasr_amd import train as tr
    from emoasr_amd.optimizers import Adam, ScheduledOptimizer
    datas = _micro_batches() + [_batch(4, [250, 240, 100]), _batch(5, [150, 140, 130, 120, 80]), _batch(6, [403])]
    params = SimpleNamespace(**CFG)

    def run(stacked):
        monkeypatch.setenv("EMOASR_STACKED", "1" if stacked else "0")
        model = _model(dev)
        opt = ScheduledOptimizer(Adam(model.parameters(), lr=0, weight_decay=params.weight_decay), params)
        calls = {"group": 0}
        orig = tr.train_group

        def counting(*a, **k):
            calls["group"] += 1
            return orig(*a, **k)

        monkeypatch.setattr(tr, "train_group", counting)
        steps = tr.train(model, opt, datas, params, dev, 0)
        monkeypatch.setattr(tr, "train_group", orig)
        assert steps == 2 and calls["group"] == (2 if stacked else 0), (steps, calls)
        return model.engine().arena.flat.clone(), opt.state_dict()["_step"]

    p_stacked, s1 = run(True)
    p_single, s2 = run(False)
    assert s1 == s2 == 2
    # Adam's first updates are sign-like (m / sqrt(v) = +-1 whatever the gradient's size), so parameters whose gradient is at
    # the noise level may move the opposite way: compare the update VECTORS, not element-wise maxima
    p0 = _model(dev).engine().arena.flat
    u1, u2 = p_stacked - p0, p_single - p0
    cos = _cos(u1, u2)
    rel = ((u1 - u2).norm() / u2.norm()).item()
    print(f"parameter updates after 2 steps, stacked vs one-by-one: cosine {cos:.5f}, relative L2 distance {rel:.3e}")
    assert u2.abs().max() > 0 and cos > 0.98 and rel < 0.2, (cos, rel)


def test_one_launch_per_kernel_equals_one_launch_per_segment(dev):
    """option "stack_launch": the per-utterance kernels (attention forward / backward, ...) take all stacked micro-batches in
    ONE launch, their segment table in the arguments (1, default), or run once per segment (0).  Same arithmetic, same dropout
    streams: losses identical, gradients equal up to the float atomics of the weight-gradient split-K."""
    from emoasr_amd import lib
    model = _model(dev, dropout_enc_rate=0.1, dropout_attn_rate=0.1)
    eng = model.engine()
    datas = _micro_batches()
    batches = [(d["xs"].to(dev), [int(v) for v in d["xlens"]], d["ys"], [int(v) for v in d["ylens"]]) for d in datas]
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}

    def run(flag):
        lib.set_option("stack_launch", flag)
        # (a per-segment attention launch of this size would split its keys over four waves -- a different order of the soft-max
        # sums than the stacked launch's; the claim here is about the launch structure, so both run unsplit)
        lib.set_option("attn_fwd_split", 0)
        try:
            model.load_state_dict(sd0)
            eng.step_count = 5
            eng.arena.grad.zero_()
            losses = eng.ctc_train_stacked(batches, 0)
            torch.cuda.synchronize()
            return losses.tolist(), eng.arena.grad.clone()
        finally:
            lib.set_option("stack_launch", 1)
            lib.set_option("attn_fwd_split", 1)

    l1, g1 = run(1)
    l0, g0 = run(0)
    assert l1 == l0, (l1, l0)
    scale = g0.abs().max().item()
    assert scale > 0 and (g1 - g0).abs().max().item() < 2e-3 * scale, (g1 - g0).abs().max().item() / scale


def test_query_pass_from_the_image_equals_the_recomputing_query_pass(dev):
    """option "attn_q2": the query pass of the attention backward reads the dS image the key pass wrote (1, default) or recomputes
    every score itself (0, the round-4 kernel) -- stacked micro-batches of different lengths (segment offsets of the image), dropout
    on.  Same keep mask, the same dS up to its bf16 rounding on the way through the image: losses identical, gradients within 5e-3
    of their scale."""
    from emoasr_amd import lib
    model = _model(dev, dropout_enc_rate=0.1, dropout_attn_rate=0.1)
    eng = model.engine()
    datas = _micro_batches()
    batches = [(d["xs"].to(dev), [int(v) for v in d["xlens"]], d["ys"], [int(v) for v in d["ylens"]]) for d in datas]
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}

    def run(flag):
        lib.set_option("attn_q2", flag)
        try:
            model.load_state_dict(sd0)
            eng.step_count = 5
            eng.arena.grad.zero_()
            losses = eng.ctc_train_stacked(batches, 0)
            torch.cuda.synchronize()
            return losses.tolist(), eng.arena.grad.clone()
        finally:
            lib.set_option("attn_q2", 1)

    l1, g1 = run(1)
    l0, g0 = run(0)
    assert l1 == l0, (l1, l0)
    assert torch.isfinite(g1).all()
    scale = g0.abs().max().item()
    assert scale > 0 and (g1 - g0).abs().max().item() < 5e-3 * scale, (g1 - g0).abs().max().item() / scale
    assert _cos(g1, g0) > 0.9999


DEC_CFGS = {
    "rnnt": dict(decoder_type="rnn_transducer", vocab_size=96, embedding_size=64, dec_hidden_size=128, dec_num_layers=2,
                 joint_hidden_size=128, dropout_emb_rate=0.0, dropout_dec_rate=0.0, mtl_ctc_weight=0.3, lsm_prob=0.0),
    "attention": dict(decoder_type="transformer", dec_hidden_size=256, dec_num_attention_heads=4, dec_num_layers=2,
                      dec_intermediate_size=512, dropout_dec_rate=0.0, mtl_ctc_weight=0.3, lsm_prob=0.1,
                      loss_normalize_length=False, loss_normalize_batch=True, max_decode_ylen=20),
}


@pytest.mark.parametrize("kind", ["rnnt", "attention"])
def test_stacked_encoder_under_other_decoders(dev, kind):
    """modeling/functions.py: encoder_apply_stacked -- the ENCODER over several micro-batches in one stacked pass, the decoder
    (RNN-T with auxiliary CTC / Transformer decoder with auxiliary CTC) per micro-batch on its slice of the stacked output --
    against the micro-batches one after the other: losses 1e-3, every parameter gradient cosine 0.999."""
    from emoasr_amd.modeling.functions import encoder_apply_stacked
    model = _model(dev, **DEC_CFGS[kind])
    eng = model.engine()
    assert eng.encoder_stacked_ok() and not eng.stacked_ok()
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    datas = _micro_batches()
    for d in datas:
        eos = torch.full((d["ys"].shape[0], 1), 2)
        d["ys_in"], d["ys_out"] = torch.cat([eos, d["ys"]], 1), torch.cat([d["ys"], eos], 1)
    n = len(datas)
    eng.arena.grad.zero_()
    want = []
    for d in datas:
        loss, _ = model(d["xs"].to(dev), d["xlens"], d["ys"], d["ylens"], d["ys_in"], d["ys_out"])
        (loss / n).backward()
        want.append(loss.item())
    want_grad = eng.arena.grad.clone()
    model.load_state_dict(sd0)
    eng.arena.grad.zero_()
    outs = encoder_apply_stacked(model.encoder, [d["xs"].to(dev) for d in datas], [d["xlens"] for d in datas])
    total, got = None, []
    for (eouts, elens, _), d in zip(outs, datas):
        loss, _, _ = model.decoder(eouts, elens, None, d["ys"], d["ylens"], d["ys_in"], d["ys_out"], None, None, None)
        total = loss / n if total is None else total + loss / n
        got.append(loss.item())
    total.backward()
    torch.cuda.synchronize()
    for a, b in zip(got, want):
        assert abs(a - b) < 1e-3 * abs(b), (got, want)
    A = eng.arena
    gmax = want_grad.abs().max().item()
    for name in A.names:
        o, k = A.offsets[name], A.pviews[name].numel()
        g, w = A.grad[o:o + k], want_grad[o:o + k]
        if w.abs().max() < 1e-3 * gmax:   # zero or rounding noise (see test_stacked_pass_equals_the_separate_passes)
            assert g.abs().max() < 2e-3 * gmax, (name, g.abs().max().item(), gmax)
            continue
        assert _cos(g, w) > 0.999, (name, _cos(g, w))


def test_train_loop_stacks_the_encoder_for_an_rnnt_model(dev, monkeypatch):
    """train.train with accum_grad 3 on an RNN-T model: stacked_ok(...) == "encoder", two optimizer steps through train_group
    (encoder of three micro-batches in one pass, decoder per micro-batch) against the one-by-one loop"""
    from emoasr_amd import train as tr
    from emoasr_amd.optimizers import Adam, ScheduledOptimizer
    datas = _micro_batches() + [_batch(4, [250, 240, 100]), _batch(5, [150, 140, 130, 120, 80]), _batch(6, [403])]
    for d in datas:
        eos = torch.full((d["ys"].shape[0], 1), 2)
        d["ys_in"], d["ys_out"] = torch.cat([eos, d["ys"]], 1), torch.cat([d["ys"], eos], 1)
    params = SimpleNamespace(**dict(CFG, **DEC_CFGS["rnnt"]))

    def run(stacked):
        monkeypatch.setenv("EMOASR_STACKED", "1" if stacked else "0")
        model = _model(dev, **DEC_CFGS["rnnt"])
        opt = ScheduledOptimizer(Adam(model.parameters(), lr=0, weight_decay=params.weight_decay), params)
        assert tr.stacked_ok(model, opt, params) == ("encoder" if stacked else False)
        steps = tr.train(model, opt, datas, params, dev, 0)
        assert steps == 2
        return model.engine().arena.flat.clone()

    p_stacked, p_single = run(True), run(False)
    p0 = _model(dev, **DEC_CFGS["rnnt"]).engine().arena.flat
    u1, u2 = p_stacked - p0, p_single - p0
    cos, rel = _cos(u1, u2), ((u1 - u2).norm() / u2.norm()).item()
    print(f"RNN-T parameter updates after 2 steps, stacked encoder vs one-by-one: cosine {cos:.5f}, relative L2 distance {rel:.3e}")
    assert u2.abs().max() > 0 and cos > 0.98 and rel < 0.2, (cos, rel)
