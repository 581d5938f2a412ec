"""Parity of configs 4 and 5 at BASELINE.json's FULL model sizes against the CPU oracle (the oracle is pinned to the
reference by the *_tiny goldens; here it is what a full-size run is compared with -- one short batch it finishes in seconds).

config 4  `L3-3`: Conformer 12x256 + Transformer decoder 6x256 (35 M), V = 10000, Transformer LM 12x256 (12 M), one
          1200-frame utterance (T' = 299), joint CTC/attention beam search: beam 10, 15 CTC candidates per beam,
          decode_ctc_weight 0.3, lm_weight 0.3 -- hypotheses identical, scores 1e-3; teacher-forced loss + gradients.
          (Random weights never end a hypothesis: <eos> is biased in the attention head, blank in the CTC head, and the LM
          head is flattened, so that ten hypotheses of 1-6 tokens finish within 7 steps.)
config 5  `L4`: Conformer 12x256 + LSTM 2x512 + joint 512 (26 M), V = 1000, three utterances (T' 100 / 90 / 75, U 13 / 12 / 10):
          transducer loss 1e-3 vs oracle.rnnt, gradient cosines, greedy ids identical.
Reference: asr/modeling/decoders/transformer.py:82-294, lm/modeling/transformer.py:62-77, rnn_transducer.py:81-240."""
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu

L2 = dict(input_layer="conv2d", feat_dim=80, num_framestacks=1, encoder_type="conformer", decoder_type="ctc",
          pos_encode_type="rel", enc_hidden_size=256, enc_num_attention_heads=4, enc_num_layers=12,
          enc_intermediate_size=1024, dropout_enc_rate=0.0, dropout_attn_rate=0.0, vocab_size=10000, blank_id=0,
          eos_id=2, kd_weight=0)
L3 = dict(L2, decoder_type="transformer", dec_hidden_size=256, dec_num_attention_heads=4, dec_num_layers=6,
          dec_intermediate_size=1024, dropout_dec_rate=0.0, mtl_ctc_weight=0.3, lsm_prob=0.1,
          loss_normalize_length=False, loss_normalize_batch=True, max_decode_ylen=12)
LM12 = dict(lm_type="transformer", vocab_size=10000, hidden_size=256, num_layers=12, num_attention_heads=4,
            intermediate_size=1024, max_seq_len=256)
L4 = dict(L2, decoder_type="rnn_transducer", vocab_size=1000, embedding_size=256, dec_hidden_size=512, dec_num_layers=2,
          joint_hidden_size=512, dropout_emb_rate=0.0, dropout_dec_rate=0.0, mtl_ctc_weight=0.3, lsm_prob=0.0)


def _cos(a, b):
    a, b = a.flatten().float(), b.flatten().float()
    return (torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)).item()


def _l3(dtype):
    from emoasr_amd.modeling.asr import ASR
    from emoasr_amd.modeling.lm import LM
    torch.manual_seed(0)
    model = ASR(SimpleNamespace(**L3), compute_dtype=dtype)
    lm = LM(SimpleNamespace(**LM12), compute_dtype=dtype)
    with torch.no_grad():
        model.decoder.output.weight.mul_(6.0)        # spread the attention head: candidates are not near-ties
        model.decoder.output.bias[2] += 8.0          # <eos> likely enough for hypotheses to end
        model.decoder.ctc.output.weight.mul_(3.0)
        model.decoder.ctc.output.bias[0] += 14.0     # blank-dominated CTC posteriors: short prefixes can cover T' frames
        for n, p in lm.named_parameters():
            if n.endswith("predictions.transform.LayerNorm.weight"):
                p.mul_(0.1)                          # a random-init LM head is wildly peaked (log-probs down to -120)
    return model, lm


def test_l3_joint_beam_search_full_size(dev):
    from oracle import decoder as od, model as om
    model, lm = _l3(torch.float32)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    lsd = {k: v.detach().clone() for k, v in lm.state_dict().items()}
    cfg = SimpleNamespace(**L3)
    g = torch.Generator().manual_seed(1)
    xs, xlens = torch.randn(1, 1200, 80, generator=g), torch.tensor([1200])
    with torch.no_grad():
        eouts, elens = om.encoder_forward(sd, cfg, xs, xlens)
        want, want_scores = od.joint_beam_search(sd, cfg, eouts, elens, 10, 0.0, (lsd, SimpleNamespace(**LM12)), 0.3, 0.3)
    assert len(want) == 10 and max(len(h) for h in want) >= 5, [len(h) for h in want]
    model, lm = model.to(dev).eval(), lm.to(dev).eval()
    hyps, scores, _, _ = model.decode(xs.to(dev), xlens, beam_width=10, len_weight=0.0, lm=lm, lm_weight=0.3,
                                      decode_ctc_weight=0.3)
    assert hyps == want, (hyps, want)
    for a, b in zip(scores, want_scores):
        assert abs(a - b) < 1e-3 * abs(b) + 2e-3, (scores, want_scores)


def test_l3_joint_beam_search_full_size_bf16_cooperative_kernels(dev):
    """The configuration bench.py times for config 4 -- bf16, beam 10, 12-layer LM, so the cached steps run as the
    cooperative launches of csrc/decode_coop.hip (one 16-workgroup launch per network, 42 / 61 grid barriers) -- against the
    f32 CPU oracle at full size, on the same sharpened weights as the f32 test above.
    bf16 contract: the SAME ten hypotheses in the same order, every score within 1 % + 0.05 of the
    oracle's (measured: 0.2 %); the cooperative path taken by both networks and none of its barriers timed out."""
    from emoasr_amd import lib
    from oracle import decoder as od, model as om
    model, lm = _l3(torch.bfloat16)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    lsd = {k: v.detach().clone() for k, v in lm.state_dict().items()}
    cfg = SimpleNamespace(**L3)
    g = torch.Generator().manual_seed(1)
    xs, xlens = torch.randn(1, 1200, 80, generator=g), torch.tensor([1200])
    with torch.no_grad():
        eouts, elens = om.encoder_forward(sd, cfg, xs, xlens)
        want, want_scores = od.joint_beam_search(sd, cfg, eouts, elens, 10, 0.0, (lsd, SimpleNamespace(**LM12)), 0.3, 0.3)
    assert len(want) == 10 and max(len(h) for h in want) >= 5, [len(h) for h in want]
    model, lm = model.to(dev).eval(), lm.to(dev).eval()
    before = [lib.size_query("emoasr_decode_coop_launches", c) for c in (0, 1)]
    hyps, scores, _, _ = model.decode(xs.to(dev), xlens, beam_width=10, len_weight=0.0, lm=lm, lm_weight=0.3,
                                      decode_ctc_weight=0.3)
    after = [lib.size_query("emoasr_decode_coop_launches", c) for c in (0, 1)]
    assert after[0] > before[0] and after[1] > before[1], ("the cooperative step kernels were not used", before, after)
    assert lib.size_query("emoasr_decode_coop_status") == 0
    print("bf16 coop search: hyps equal", hyps == want, "max rel score err",
          max(abs(a - b) / abs(b) for a, b in zip(scores, want_scores)) if len(scores) == len(want_scores) else None)
    assert hyps == want, (hyps, want)
    for a, b in zip(scores, want_scores):
        assert abs(a - b) < 1e-2 * abs(b) + 5e-2, (scores, want_scores)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_l3_teacher_forced_loss_and_grads_full_size(dev, dtype):
    from oracle import decoder as od, model as om
    model, _ = _l3(dtype)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cfg = SimpleNamespace(**L3)
    g = torch.Generator().manual_seed(2)
    xlens, ylens = torch.tensor([403, 367, 298]), torch.tensor([13, 12, 9])
    xs = torch.randn(3, 403, 80, generator=g)
    ys = torch.randint(3, 10000, (3, 13), generator=g)
    for b in range(3):
        xs[b, xlens[b]:] = 0
        ys[b, ylens[b]:] = 2
    eos = torch.full((3, 1), 2)
    ys_in, ys_out = torch.cat([eos, ys], 1), torch.cat([ys, eos], 1)
    params = {k: v for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    for v in params.values():
        v.requires_grad_(True)
    eouts, elens = om.encoder_forward(sd, cfg, xs, xlens, training=True)
    loss_ref, ld_ref, _ = od.decoder_forward(sd, cfg, eouts, elens, ys, ylens, ys_in, ys_out)
    loss_ref.backward()
    model = model.to(dev).train()
    loss, ld = model(xs.to(dev), xlens, ys, ylens, ys_in, ys_out)
    loss.backward()
    ltol = 1e-3 if dtype == torch.float32 else 2e-2
    for k in ("loss_att", "loss_ctc", "loss_total"):
        assert abs(ld[k].item() - ld_ref[k].item()) < ltol * abs(ld_ref[k].item()), (k, ld[k].item(), ld_ref[k].item())
    grads = {n: p.grad.float().cpu() for n, p in model.named_parameters()}
    for name in ("decoder.output.weight", "decoder.embed.weight", "decoder.transformers.5.src_attn.linear_k.weight",
                 "decoder.transformers.0.self_attn.linear_q.weight", "decoder.transformers.2.feed_forward.w1.weight",
                 "decoder.ctc.output.weight", "encoder.transformers.11.feed_forward.w2.weight",
                 "encoder.transformers.4.self_attn.pos_bias_v", "encoder.conv.conv.2.weight"):
        cos = _cos(grads[name], params[name].grad)
        assert cos > (0.9995 if dtype == torch.float32 else 0.97), (name, cos)


@pytest.mark.parametrize("dtype", [torch.float32, "f32x3", torch.bfloat16], ids=["f32", "f32x3", "bf16"])
def test_l4_transducer_full_size(dev, dtype):
    """the transducer gradient has no reference-held pin (the reference calls warp_rnnt): at the full model size it is held to
    the oracle's autograd through the path-enumeration-validated lattice restatement (oracle/rnnt.py) -- f32 / f32x3: every
    sampled parameter gradient cosine >= 0.9999 (printed)"""
    from emoasr_amd.modeling.asr import ASR
    from oracle import model as om, rnnt as orn
    torch.manual_seed(0)
    mode = dtype
    model = ASR(SimpleNamespace(**L4), compute_dtype=dtype)
    dtype = torch.float32 if mode == "f32x3" else dtype
    with torch.no_grad():
        # random LSTM / joint weights give a joint output that hardly depends on the label history: greedy decoding then
        # repeats one label until the 256-symbol cap.  Louder embedding / projections and a blank bias make it a mixed
        # sequence of blanks and several labels (it still runs into the cap: 257 joint evaluations per utterance compared)
        model.decoder.embed.weight.mul_(20.0)
        model.decoder.w_dec.weight.mul_(20.0)
        model.decoder.w_enc.weight.mul_(6.0)
        model.decoder.output.weight.mul_(2.0)
        model.decoder.output.bias[0] += 5.5
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cfg = SimpleNamespace(**L4)
    g = torch.Generator().manual_seed(3)
    xlens, ylens = torch.tensor([403, 363, 303]), torch.tensor([13, 12, 10])
    xs = torch.randn(3, 403, 80, generator=g)
    ys = torch.randint(3, 1000, (3, 13), generator=g)
    for b in range(3):
        xs[b, xlens[b]:] = 0
        ys[b, ylens[b]:] = 2
    eos = torch.full((3, 1), 2)
    ys_in, ys_out = torch.cat([eos, ys], 1), torch.cat([ys, eos], 1)
    params = {k: v for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    for v in params.values():
        v.requires_grad_(True)
    eouts, elens = om.encoder_forward(sd, cfg, xs, xlens, training=True)
    assert [int(e) for e in elens] == [100, 90, 75]
    loss_ref, ld_ref, _ = orn.rnnt_decoder_forward(sd, cfg, eouts, elens, ys, ylens, ys_in)
    loss_ref.backward()
    model = model.to(dev).train()
    loss, ld = model(xs.to(dev), xlens, ys, ylens, ys_in, ys_out)
    loss.backward()
    ltol = 1e-3 if dtype == torch.float32 else 2e-2
    for k in ("loss_rnnt", "loss_ctc", "loss_total"):
        assert abs(ld[k].item() - ld_ref[k].item()) < ltol * abs(ld_ref[k].item()), (k, ld[k].item(), ld_ref[k].item())
    grads = {n: p.grad.float().cpu() for n, p in model.named_parameters()}
    names = [n for n in grads if n.startswith("decoder.")][:12] + ["encoder.transformers.11.feed_forward.w2.weight",
                                                                    "encoder.conv.conv.2.weight"]
    for name in names:
        if params[name].grad.abs().max() < 1e-8:
            continue
        cos = _cos(grads[name], params[name].grad)
        print(f"[measured L4 {mode}] grad cosine {name}: {cos:.6f}")
        assert cos > (0.9999 if dtype == torch.float32 else 0.97), (name, cos)
    # ---- greedy decoding, eval mode from the ORIGINAL state (the training step above moved the BatchNorm running stats)
    model.load_state_dict({k: v.detach() for k, v in sd.items()})
    model.eval()
    with torch.no_grad():
        sd_eval = {k: v.detach() for k, v in sd.items()}
        e2, el2 = om.encoder_forward(sd_eval, cfg, xs, xlens)
        want, _ = orn.rnnt_greedy(sd_eval, cfg, e2, el2)
    hyps = model.decode(xs.to(dev), xlens)[0]
    assert sum(len(h) for h in want) > 0
    if dtype == torch.float32:
        assert hyps == want, (hyps, want)
    else:
        # agreement by ALIGNMENT (1 - edit distance / reference length, emoasr_amd.metrics): a position-by-position comparison of
        # the label sequences counts every label behind one inserted / dropped label as wrong -- on these random-init weights
        # (near-ties over the vocabulary) it moved between 0.75 and 0.9 with the rounding of one attention kernel
        from emoasr_amd.metrics import compute_wer
        errs = sum(compute_wer([str(t) for t in h], [str(t) for t in w])[1]["wer"] * len(w) / 100.0 for h, w in zip(hyps, want) if len(w))
        agree = 1.0 - errs / max(1, sum(len(w) for w in want))
        print(f"[measured L4 {dtype}] greedy label agreement (1 - edit distance / length) {agree:.4f}")
        assert agree > 0.8, agree
