"""Parity at BASELINE.json's full model size (L2: 23.5 M Conformer-CTC, d 256, 12 layers, V 10000).

1. Against the oracle directly: a three-utterance batch is small enough for the CPU restatement to finish in
   seconds at the FULL model size -- loss, logits, greedy ids, gradients (f32: 1e-3 as north_star states,
   greedy ids bit-exact; bf16: loss 2e-3, logits 3e-2 of range, gradient cosines 0.995, greedy agreement 0.9 -- the
   measured values, printed by the test, are 2.0e-4 / 1.25e-2 / >= 0.999 / 0.957).
2. Size-independent properties on a full bench-sized batch (about 27 k frames, bf16):
   CTC logit gradients sum to zero over the vocabulary on valid frames and vanish on padded ones;
   the backward pass is linear in the incoming loss gradient; eval-mode decoding of an equal-length batch
   equals decoding each utterance alone; the counter-based dropout is reproducible for a given step."""
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu

L2 = dict(input_layer="conv2d", feat_dim=80, num_framestacks=1, encoder_type="conformer", decoder_type="ctc",
          pos_encode_type="rel", enc_hidden_size=256, enc_num_attention_heads=4, enc_num_layers=12,
          enc_intermediate_size=1024, dropout_enc_rate=0.0, dropout_attn_rate=0.0, vocab_size=10000, blank_id=0,
          eos_id=2, kd_weight=0)


def _model(dtype, dev, seed=0, **over):
    from emoasr_amd.modeling.asr import ASR
    torch.manual_seed(seed)
    model = ASR(SimpleNamespace(**dict(L2, **over)), compute_dtype=dtype)
    with torch.no_grad():  # spread the head so greedy decoding is not all blanks / ties
        model.decoder.output.weight.mul_(3.0)
        for n, p in model.named_parameters():
            if "batch_norm" in n or ".norm" in n:
                p.add_(0.05 * torch.randn_like(p))
    return model.to(dev)


def _batch(seed, xlens, V=10000):
    g = torch.Generator().manual_seed(seed)
    xlens = torch.tensor(xlens)
    ylens = torch.clamp(xlens // 30, min=1)
    B, T, L = len(xlens), int(xlens.max()), int(ylens.max())
    xs = torch.randn(B, T, 80, generator=g)
    ys = torch.randint(3, V, (B, L), generator=g)
    for b in range(B):
        xs[b, xlens[b]:] = 0
        ys[b, ylens[b]:] = 2
    return xs, xlens, ys, ylens


@pytest.mark.parametrize("dtype", [torch.float32, "f32x3", torch.bfloat16], ids=["f32", "f32x3", "bf16"])
def test_full_model_against_oracle(dev, dtype):
    """"f32x3" (f32 storage, every matrix product as three bf16 MFMAs over (hi, lo) operand pairs) is held to the f32 bars:
    loss / logits 1e-3, gradient cosines 0.9995, greedy ids bit-exact on these random-init weights"""
    from oracle import model as om
    mode = dtype
    model = _model(dtype, dev)
    dtype = torch.float32 if mode == "f32x3" else dtype
    assert model.compute_dtype == dtype and model.f32_split == (mode == "f32x3")
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    cfg = SimpleNamespace(**L2)
    xs, xlens, ys, ylens = _batch(1, [403, 367, 298])
    # ---- train mode (BatchNorm batch statistics), loss + gradients
    params = {k: v for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    for v in params.values():
        v.requires_grad_(True)
    loss_ref, _, logits_ref = om.asr_ctc_forward(sd, cfg, xs, xlens, ys, ylens, training=True)
    loss_ref.backward()
    model.train()
    loss, ld = model(xs.to(dev), xlens, ys, ylens, None, None)
    loss.backward()
    ltol = 1e-3 if dtype == torch.float32 else 2e-3  # bf16 measured 2.0e-4 (round 2)
    print(f"[measured {mode}] loss rel err {abs(loss.item() - loss_ref.item()) / abs(loss_ref.item()):.2e}")
    assert abs(loss.item() - loss_ref.item()) < ltol * abs(loss_ref.item()), (loss.item(), loss_ref.item())
    grads = {n: p.grad.float().cpu() for n, p in model.named_parameters()}
    for name in ("decoder.output.weight", "encoder.norm.weight", "encoder.transformers.11.feed_forward.w2.weight",
                 "encoder.transformers.6.self_attn.linear_pos.weight", "encoder.transformers.0.conv.depthwise_conv.weight",
                 "encoder.transformers.3.self_attn.pos_bias_u", "encoder.conv.conv.2.weight", "encoder.conv.conv.0.weight"):
        a, b = grads[name].flatten(), params[name].grad.flatten()
        cos = (torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)).item()
        print(f"[measured {mode}] grad cosine {name}: {cos:.5f}")
        assert cos > (0.9995 if dtype == torch.float32 else 0.995), (name, cos)  # bf16 measured >= 0.9990
        if dtype == torch.float32:
            assert abs(a.norm().item() / b.norm().item() - 1) < 5e-3, name
    # ---- eval mode: logits + greedy ids (the reference's running statistics are untouched in `sd`)
    with torch.no_grad():
        sd_eval = {k: v.detach() for k, v in sd.items()}
        eouts, elens = om.encoder_forward(sd_eval, cfg, xs, xlens)
        logits_ref = om.ctc_decoder_forward(sd_eval, cfg, eouts, elens)
        want, want_aligns = om.ctc_greedy(logits_ref, elens, 0)
    model.load_state_dict({k: v.detach() for k, v in sd.items()})  # undo the BatchNorm running-stat update
    model.eval()
    with torch.no_grad():
        e2, el2, _ = model.encoder(xs.to(dev), xlens)
        logits = model.decoder(e2, el2)
    hyps, _, _, aligns = model.decode(xs.to(dev), xlens)
    rel = ((logits.float().cpu() - logits_ref).abs().max() / logits_ref.abs().max()).item()
    print(f"[measured {mode}] logits rel err {rel:.2e}")
    assert rel < (1e-3 if dtype == torch.float32 else 3e-2), rel  # bf16 measured 1.25e-2 of the logits' range
    if dtype == torch.float32:
        assert hyps == want
    else:
        # frame-level arg-max agreement (random-init weights: near-ties over V = 10 000; a token-by-token comparison of the
        # collapsed hypotheses shifts at every flipped frame -- 0.957 / 0.827 measured that way over two library builds with the
        # same 1.26e-2 logits error)
        same = sum(int(a == b) for ga, wa in zip(aligns, want_aligns) for a, b in zip(ga, wa))
        agree = same / max(1, sum(len(wa) for wa in want_aligns))
        print(f"[measured {dtype}] greedy frame agreement {agree:.4f}")
        assert agree > 0.9, agree


_BENCH_ORACLE = {}


def _bench_batch_oracle():
    """one sampler-packed micro-batch of the benchmark (bench.make_batches: 30 000 frames / <= 50 utterances of the synthetic
    LibriSpeech-shaped set) through oracle/model.py on the host, once per test session: loss, every gradient, eval logits, greedy"""
    if _BENCH_ORACLE:
        return _BENCH_ORACLE
    import bench
    from oracle import model as om
    batch = bench.make_batches(0, 1, 1, torch.device("cpu"))[0]
    xs, xlens, ys, ylens = batch.xs, torch.tensor(batch.xlens), batch.ys, torch.tensor(batch.ylens)
    model = _model(torch.float32, torch.device("cpu"))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cfg = SimpleNamespace(**L2)
    params = {k: v for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    for v in params.values():
        v.requires_grad_(True)
    loss_ref, _, _ = om.asr_ctc_forward(sd, cfg, xs, xlens, ys, ylens, training=True)
    loss_ref.backward()
    with torch.no_grad():
        sd_eval = {k: v.detach() for k, v in sd.items()}
        eouts, elens = om.encoder_forward(sd_eval, cfg, xs, xlens)
        logits_ref = om.ctc_decoder_forward(sd_eval, cfg, eouts, elens)
        want, want_aligns = om.ctc_greedy(logits_ref, elens, 0)
    _BENCH_ORACLE.update(xs=xs, xlens=xlens, ys=ys, ylens=ylens, sd=sd_eval, loss=loss_ref.item(),
                         grads={k: v.grad.clone() for k, v in params.items()}, logits=logits_ref, hyps=want, aligns=want_aligns)
    return _BENCH_ORACLE


@pytest.mark.parametrize("mode", ["f32x3", "bf16"])
def test_bench_sized_batch_against_oracle(dev, mode):
    """Oracle parity AT THE BENCH BATCH (round 6; rounds 1-5 checked this size through properties only): the benchmark's own first
    micro-batch (~30 k input frames, 36 utterances, padded lengths as the sampler packs them) at the full model size, dropout 0,
    against oracle/model.py on the host.  f32x3: loss 1e-3, logits 1e-3 of range, greedy ids bit-exact, gradient cosines 0.9995;
    bf16 (the benched mode): loss 2e-3, logits 3e-2 of range, frame agreement 0.9, gradient cosines 0.995.
    Reference: asr/modeling/asr.py:53-68."""
    o = _bench_batch_oracle()
    dtype = torch.float32 if mode == "f32x3" else torch.bfloat16
    model = _model("f32x3" if mode == "f32x3" else dtype, dev)
    model.load_state_dict(o["sd"])
    xs, xlens, ys, ylens = o["xs"].to(dev), o["xlens"], o["ys"], o["ylens"]
    model.train()
    loss, _ = model(xs, [int(v) for v in xlens], ys, [int(v) for v in ylens], None, None)
    loss.backward()
    rel = abs(loss.item() - o["loss"]) / abs(o["loss"])
    print(f"[measured {mode}, {int(xlens.sum())} frames] loss rel err {rel:.2e}")
    assert rel < (1e-3 if mode == "f32x3" else 2e-3), (loss.item(), o["loss"])
    grads = {n: p.grad.float().cpu() for n, p in model.named_parameters()}
    for name in ("decoder.output.weight", "encoder.norm.weight", "encoder.transformers.11.feed_forward.w2.weight",
                 "encoder.transformers.6.self_attn.linear_pos.weight", "encoder.transformers.0.conv.depthwise_conv.weight",
                 "encoder.transformers.3.self_attn.pos_bias_u", "encoder.conv.conv.2.weight", "encoder.conv.conv.0.weight"):
        a, b = grads[name].flatten(), o["grads"][name].flatten()
        cos = (torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)).item()
        print(f"[measured {mode}] grad cosine {name}: {cos:.5f}")
        assert cos > (0.9995 if mode == "f32x3" else 0.995), (name, cos)
    model.load_state_dict(o["sd"])   # undo the BatchNorm running-stat update
    model.eval()
    with torch.no_grad():
        e2, el2, _ = model.encoder(xs, xlens)
        logits = model.decoder(e2, el2)
    hyps, _, _, aligns = model.decode(xs, xlens)
    relg = ((logits.float().cpu() - o["logits"]).abs().max() / o["logits"].abs().max()).item()
    print(f"[measured {mode}] logits rel err {relg:.2e}")
    assert relg < (1e-3 if mode == "f32x3" else 3e-2), relg
    if mode == "f32x3":
        assert hyps == o["hyps"]
    else:
        same = sum(int(a == b) for ga, wa in zip(aligns, o["aligns"]) for a, b in zip(ga, wa))
        agree = same / max(1, sum(len(wa) for wa in o["aligns"]))
        print(f"[measured bf16] greedy frame agreement {agree:.4f}")
        assert agree > 0.9, agree


def test_bf16_greedy_hypotheses_exact_on_fitted_weights(dev):
    """bf16 decoding contract: on TRAINED-like weights bf16 greedy hypotheses are IDENTICAL to the f32
    CPU oracle's.  Random-init weights give near-uniform posteriors over V = 10000 whose arg-max flips under bf16 rounding
    (frame agreement 0.96 above); a trained model's are peaked.  The full-size model is therefore first fitted -- by the HIP
    training path itself, bf16, Adam, up to 300 updates on one three-utterance batch -- until its CTC loss per utterance is
    below 0.1 (and for at least 160 updates, so that the BatchNorm running statistics have converged), then the fitted state is decoded by the bf16 engine and by oracle/model.py (f32, CPU) from the same state dict.
    Reference: decoders/ctc.py:176-201 (_greedy)."""
    from emoasr_amd.optimizers import Adam
    from oracle import model as om
    model = _model(torch.bfloat16, dev)
    cfg = SimpleNamespace(**L2)
    xs, xlens, ys, ylens = _batch(11, [403, 367, 298])
    opt = Adam(model.parameters(), lr=5e-4)
    opt.clip_grad_norm = 5.0
    model.train()
    xd = xs.to(dev)
    last = None
    for it in range(400):
        loss, _ = model(xd, xlens, ys, ylens, None, None)
        loss.backward()
        opt.step()
        opt.zero_grad()
        if it % 20 == 19:
            last = loss.item()
            # (at least 160 updates: the BatchNorm running statistics that eval-mode decoding uses follow the batch statistics
            # with momentum 0.1 -- the loss itself is below 0.1 after 20 updates)
            if last < 0.1 and it >= 159:
                break
    print(f"[fitted] {it + 1} updates, CTC loss {last:.3f}")
    assert last is not None and last < 0.1, last
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        eouts, elens = om.encoder_forward(sd, cfg, xs, xlens)
        logits_ref = om.ctc_decoder_forward(sd, cfg, eouts, elens)
        want, _ = om.ctc_greedy(logits_ref, elens, 0)
    model.eval()
    hyps, _, logits, aligns = model.decode(xd, xlens)
    assert sum(len(h) for h in want) >= 20, want          # a real transcription, not blanks
    assert hyps == want, (hyps, want)
    labels = [[int(t) for t in ys[b, :ylens[b]]] for b in range(3)]
    print(f"[fitted] hypotheses equal the training labels: {hyps == labels}")
    # frame level: the arg-max path itself (before collapsing) agrees on every valid frame
    ref_path = logits_ref.argmax(-1)
    for b in range(3):
        assert [int(v) for v in ref_path[b, :int(elens[b])]] == [int(v) for v in aligns[b]], b


def _bench_batch(seed=3):
    # LibriSpeech-shaped, sorted by length, packed to ~27 k frames like the sampler does
    g = torch.Generator().manual_seed(seed)
    xlens = sorted(int(v) for v in torch.randint(1180, 1420, (21,), generator=g))
    return _batch(seed, xlens)


def test_ctc_gradient_rows_full_vocabulary(dev):
    from emoasr_amd import ops
    xs, xlens, ys, ylens = _bench_batch()
    B = len(xlens)
    elens = ((xlens - 1) // 2 - 1) // 2
    T = int(elens.max())
    logits = (2.0 * torch.randn(B, T, 10000, generator=torch.Generator().manual_seed(0))).to(torch.bfloat16).to(dev)
    i32 = lambda t: t.to(torch.int32).to(dev)
    lse = ops.row_lse(logits.view(B * T, -1))
    lp, alpha, beta, nll = ops.ctc_forward(logits, lse, i32(ys), i32(elens), i32(ylens), 0)
    assert torch.isfinite(nll).all() and (nll > 0).all()
    grad = ops.ctc_grad(logits, lse, i32(ys), i32(elens), i32(ylens), 0, lp, alpha, beta, nll, 1.0 / B).float()
    rows = grad.sum(-1)
    valid = torch.arange(T, device=dev)[None, :] < elens.to(dev)[:, None]
    # softmax minus occupancy: both sum to one over the vocabulary (bf16 rounding of 10 k addends remains)
    assert rows[valid].abs().max() < 2e-3 / B * 50, rows[valid].abs().max()
    assert grad[~valid].abs().max() == 0
    # the label occupancies of one utterance sum to its label count ... per frame: sum over non-blank = 1 - blank occupancy
    occ = torch.softmax(logits.float(), -1) / B - grad
    assert torch.allclose(occ[valid].sum(-1), torch.full_like(occ[valid].sum(-1), 1.0 / B), atol=2e-3 / B * 50)


def test_backward_is_linear_in_the_loss_gradient(dev):
    model = _model(torch.bfloat16, dev, dropout_enc_rate=0.1, dropout_attn_rate=0.1)
    model.train()
    eng = model.engine()
    xs, xlens, ys, ylens = _bench_batch()
    xs = xs.to(dev)

    def grads(scale):
        eng.step_count = 7  # same dropout masks for every run
        model.zero_grad(set_to_none=False)
        loss, _ = model(xs, xlens, ys, ylens, None, None)
        (loss * scale).backward()
        return loss.item(), eng.arena.grad.clone()

    l1, g1 = grads(1.0)
    l1b, g1b = grads(1.0)
    l2, g2 = grads(-2.5)
    assert l1 == l1b == l2  # counter-based dropout: the forward pass is reproducible bit for bit
    # (weight gradients accumulate with float atomics: two identical runs differ in the last bits)
    scale = g1.abs().max().item()
    assert (g1 - g1b).abs().max().item() < 2e-3 * scale
    assert (g2 + 2.5 * g1).abs().max().item() < 6e-3 * scale * 2.5
    assert torch.isfinite(g1).all() and g1.abs().max() > 0


def test_equal_length_batch_decodes_like_single_utterances(dev):
    model = _model(torch.bfloat16, dev)
    model.eval()
    xs, xlens, ys, ylens = _batch(5, [1217] * 6)
    hyps, _, logits, aligns = model.decode(xs.to(dev), xlens)
    for b in range(0, 6, 2):
        h1, _, lg1, a1 = model.decode(xs[b:b + 1].to(dev), xlens[b:b + 1])
        rel = ((lg1[0] - logits[b]).abs().max() / logits[b].abs().max()).item()
        assert rel < 2e-2, rel  # different tile shapes -> different bf16 summation order
        agree = sum(int(x == y) for x, y in zip(a1[0], aligns[b])) / len(aligns[b])
        assert agree > 0.97, agree


L1 = dict(input_layer="conv2d", feat_dim=80, num_framestacks=1, encoder_type="transformer", decoder_type="ctc",
          enc_hidden_size=256, enc_num_attention_heads=4, enc_num_layers=12, enc_intermediate_size=2048,
          dropout_enc_rate=0.0, dropout_attn_rate=0.0, vocab_size=10000, blank_id=0, eos_id=2, kd_weight=0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_l1_transformer_ctc_full_size_against_oracle(dev, dtype):
    """BASELINE.json config 1 (`L1`: CTC(Transformer), 20.19 M parameters, the reference's CPU-runnable plumbing case: 4 utterances
    of 1200 / 1037 / 911 / 640 frames) on the HIP path at its FULL size against the oracle: absolute
    positions, pre-norm Transformer layers with LayerNorm eps 1e-12 and ReLU feed-forward blocks of 2048 (asr/modeling/
    transformer.py:121-153, encoders/transformer.py:84-113).  f32: loss 1e-3, logits 1e-3, greedy ids identical, gradient cosines
    0.9995; bf16: the bars of the L2 test above."""
    from emoasr_amd.modeling.asr import ASR
    from oracle import model as om
    torch.manual_seed(0)
    model = ASR(SimpleNamespace(**L1), compute_dtype=dtype)
    assert abs(sum(p.numel() for p in model.parameters()) / 1e6 - 20.19) < 0.01
    with torch.no_grad():
        model.decoder.output.weight.mul_(3.0)
        for n, p in model.named_parameters():
            if ".norm" in n:
                p.add_(0.05 * torch.randn_like(p))
    model = model.to(dev)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    cfg = SimpleNamespace(**L1)
    xs, xlens, ys, ylens = _batch(7, [1200, 1037, 911, 640])
    params = {k: v for k, v in sd.items() if v.dtype.is_floating_point}
    for v in params.values():
        v.requires_grad_(True)
    loss_ref, _, _ = om.asr_ctc_forward(sd, cfg, xs, xlens, ys, ylens, training=True)
    loss_ref.backward()
    model.train()
    loss, _ = model(xs.to(dev), xlens, ys, ylens, None, None)
    loss.backward()
    rel = abs(loss.item() - loss_ref.item()) / abs(loss_ref.item())
    print(f"[measured L1 {dtype}] loss rel err {rel:.2e}")
    assert rel < (1e-3 if dtype == torch.float32 else 2e-3)
    grads = {n: p.grad.float().cpu() for n, p in model.named_parameters()}
    worst = 1.0
    for name in ("decoder.output.weight", "encoder.norm.weight", "encoder.transformers.11.feed_forward.w2.weight",
                 "encoder.transformers.6.self_attn.linear_q.weight", "encoder.transformers.0.feed_forward.w1.bias",
                 "encoder.transformers.3.norm1.weight", "encoder.conv.conv.2.weight", "encoder.conv.output.weight"):
        a, b = grads[name].flatten(), params[name].grad.flatten()
        cos = (torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)).item()
        worst = min(worst, cos)
        assert cos > (0.9995 if dtype == torch.float32 else 0.995), (name, cos)
    print(f"[measured L1 {dtype}] worst sampled gradient cosine {worst:.5f}")
    with torch.no_grad():
        sd_eval = {k: v.detach() for k, v in sd.items()}
        eouts, elens = om.encoder_forward(sd_eval, cfg, xs, xlens)
        logits_ref = om.ctc_decoder_forward(sd_eval, cfg, eouts, elens)
        want, want_aligns = om.ctc_greedy(logits_ref, elens, 0)
    model.eval()
    with torch.no_grad():
        e2, el2, _ = model.encoder(xs.to(dev), xlens)
        logits = model.decoder(e2, el2)
    hyps, _, _, aligns = model.decode(xs.to(dev), xlens)
    lrel = ((logits.float().cpu() - logits_ref).abs().max() / logits_ref.abs().max()).item()
    print(f"[measured L1 {dtype}] logits rel err {lrel:.2e}")
    assert lrel < (1e-3 if dtype == torch.float32 else 3e-2), lrel
    if dtype == torch.float32:
        assert hyps == want
    else:
        # frame-level arg-max agreement (on random-init weights the ~200-token hypotheses differ by insertions, which shift every
        # later position of a token-by-token comparison: 0.64 measured that way)
        same = sum(int(a == b) for ga, wa in zip(aligns, want_aligns) for a, b in zip(ga, wa))
        agree = same / max(1, sum(len(wa) for wa in want_aligns))
        print(f"[measured L1 {dtype}] greedy frame agreement {agree:.4f}")
        assert agree > 0.9, agree
