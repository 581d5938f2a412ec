"""autograd glue: two Functions connect the engine's explicit forward/backward to
`loss.backward()`, one for the encoder stack and one for the CTC head, so the
encoder/decoder module boundary of the reference survives.

Parameter gradients are accumulated by the engine directly into the flat gradient arena
(the parameters' .grad tensors are views of it); the Functions therefore return None for
the parameter inputs.
"""
import torch


def _engine_of(mod):
    if mod._owner is None:
        raise RuntimeError("emoasr_amd: encoder/decoder modules run through their owning ASR model")
    return mod._owner[0].engine()


def _host_list(lens):
    if torch.is_tensor(lens):
        return lens.tolist()  # device tensors sync here, like max(xlens) in asr.py:57
    return [int(v) for v in lens]


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, training, xs, xlens_host, *params):
        eouts, elens_host, elens_dev, st = eng.forward(xs, xlens_host, training, stash=True)
        ctx.eng, ctx.st = eng, st
        ctx.mark_non_differentiable(elens_dev)
        return eouts, elens_dev

    @staticmethod
    def backward(ctx, deouts, _):
        ctx.eng.backward(ctx.st, deouts.contiguous())
        ctx.st = None
        return (None,) * (4 + len(ctx.eng.arena.params))


def encoder_apply(enc, xs, xlens):
    eng = _engine_of(enc)
    xs = xs.to(torch.float32).contiguous()
    host = _host_list(xlens)
    if torch.is_grad_enabled():
        eng.step_count += 1
        eouts, elens_dev = _EncoderFn.apply(eng, enc.training, xs, host, *eng.arena.params)
    else:
        eouts, _, elens_dev, _ = eng.forward(xs, host, enc.training, stash=False)
    elens = torch.tensor([((v - 1) // 2 - 1) // 2 for v in host], dtype=torch.int64)
    if torch.is_tensor(xlens):
        elens = elens.to(xlens.device)
    eouts._emo_elens_dev = elens_dev
    return eouts, elens


class _CTCLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, eouts, elens_dev, ys_host, ylens_host, blank, *params):
        logits = eng.head_logits(eouts)
        need = eouts.requires_grad or any(p.requires_grad for p in params)
        loss, cctx = eng.ctc_loss(logits, elens_dev, ys_host, ylens_host, blank, need)
        ctx.eng, ctx.cctx, ctx.eouts = eng, cctx, eouts
        ctx.mark_non_differentiable(logits)
        return loss, logits

    @staticmethod
    def backward(ctx, gloss, _):
        eng = ctx.eng
        gdev = gloss.to(torch.float32).reshape(1) if gloss.is_cuda else None  # no host sync
        dlogits = eng.ctc_grad(ctx.cctx, 1.0 if gdev is not None else float(gloss), gdev)
        deouts = eng.head_backward(ctx.eouts, dlogits)
        ctx.cctx = None
        return (None, deouts, None, None, None, None) + (None,) * len(eng.arena.params)


def _elens_dev(eouts, elens):
    dev = getattr(eouts, "_emo_elens_dev", None)
    if dev is None:
        dev = torch.as_tensor(elens).to(torch.int32).to(eouts.device)
    return dev


def ctc_head_apply(dec, eouts):
    return _engine_of(dec).head_logits(eouts)


def ctc_loss_apply(dec, eouts, elens, ys, ylens):
    eng = _engine_of(dec)
    ys_host = ys.cpu() if torch.is_tensor(ys) else torch.as_tensor(ys)
    return _CTCLossFn.apply(eng, eouts, _elens_dev(eouts, elens), ys_host, _host_list(ylens), dec.blank_id,
                            *eng.arena.params)


def ctc_greedy_apply(dec, eouts, elens):
    eng = _engine_of(dec)
    logits = eng.head_logits(eouts)
    best, hyp, hyplen = eng.greedy(logits, _elens_dev(eouts, elens), dec.blank_id)
    best_h, hyp_h, n_h = best.cpu(), hyp.cpu(), hyplen.cpu().tolist()  # one D2H per batch
    el = _host_list(elens)
    hyps = [hyp_h[b, : n_h[b]].tolist() for b in range(len(n_h))]
    aligns = [best_h[b, : el[b]].tolist() for b in range(len(n_h))]
    return hyps, [None] * len(hyps), logits, aligns


# ---------------------------------------------------------------------------------------
# Transformer decoder: attention loss (+ auxiliary CTC) as one autograd node over the encoder output
# ---------------------------------------------------------------------------------------
class _AttnDecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, training, eouts, elens_dev, ys_host, ylens_host, ys_in, ys_out, blank, *params):
        logits, st = eng.dec_forward(eouts, elens_dev, ys_in, ylens_host, training, True)
        loss_att, _ = eng.att_loss(logits, ys_out, ylens_host)
        loss_ctc, cctx, loss = None, None, loss_att
        if eng.mtl_ctc > 0:
            ctc_logits = eng.head_logits(eouts, "decoder.ctc.output")
            loss_ctc, cctx = eng.ctc_loss(ctc_logits, elens_dev, ys_host, ylens_host, blank, True)
            loss = loss_att + eng.mtl_ctc * loss_ctc
        else:
            loss_ctc = torch.zeros_like(loss_att)
        ctx.eng, ctx.st, ctx.cctx, ctx.eouts, ctx.logits = eng, st, cctx, eouts, logits
        ctx.ys_out, ctx.ylens_host = ys_out, ylens_host
        ctx.mark_non_differentiable(logits)
        return loss, loss_att, loss_ctc, logits

    @staticmethod
    def backward(ctx, g_total, g_att, g_ctc, _):
        eng = ctx.eng
        g_att_eff = (g_total + g_att).to(torch.float32).reshape(1)
        _, dlogits = eng.att_loss(ctx.logits, ctx.ys_out, ctx.ylens_host, True, g_att_eff)
        deouts = eng.dec_backward(ctx.st, dlogits)
        if ctx.cctx is not None:
            g_ctc_eff = (g_total * eng.mtl_ctc + g_ctc).to(torch.float32).reshape(1)
            dcl = eng.ctc_grad(ctx.cctx, 1.0, g_ctc_eff)
            from .. import ops
            deouts = ops.add(deouts, eng.head_backward(ctx.eouts, dcl, "decoder.ctc.output"))
        ctx.st = ctx.cctx = None
        return (None, None, deouts, None, None, None, None, None, None) + (None,) * len(eng.arena.params)


def attn_decoder_apply(dec, eouts, elens, ys, ylens, ys_in, ys_out):
    eng = _engine_of(dec)
    ylens_host = _host_list(ylens)
    L = max(ylens_host) + 1
    ys_host = ys.cpu() if torch.is_tensor(ys) else torch.as_tensor(ys)
    ys_in = (ys_in.cpu() if torch.is_tensor(ys_in) else torch.as_tensor(ys_in))[:, :L]
    ys_out = (ys_out.cpu() if torch.is_tensor(ys_out) else torch.as_tensor(ys_out))[:, :L]
    return _AttnDecoderFn.apply(eng, dec.training, eouts, _elens_dev(eouts, elens), ys_host, ylens_host, ys_in, ys_out,
                                dec.blank_id, *eng.arena.params)


def attn_decoder_logits(dec, eouts, elens, ys_in, ylens):
    eng = _engine_of(dec)
    ys_in = ys_in.cpu() if torch.is_tensor(ys_in) else torch.as_tensor(ys_in)
    ylens_host = _host_list(ylens) if ylens is not None else [ys_in.shape[1] - 1] * ys_in.shape[0]
    with torch.no_grad():
        logits, _ = eng.dec_forward(eouts, _elens_dev(eouts, elens), ys_in, ylens_host, dec.training, False)
    return logits


# ---------------------------------------------------------------------------------------
# RNN-T: transducer loss (+ auxiliary CTC) as one autograd node over the encoder output
# ---------------------------------------------------------------------------------------
class _RNNTFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, training, eouts, elens_dev, ys_host, ylens_host, ys_in, blank, *params):
        loss_rnnt, logits, st = eng.rnnt_forward(eouts, elens_dev, ys_in, ys_host, ylens_host, blank, training)
        cctx, loss = None, loss_rnnt
        if eng.mtl_ctc > 0:
            ctc_logits = eng.head_logits(eouts, "decoder.ctc.output")
            loss_ctc, cctx = eng.ctc_loss(ctc_logits, elens_dev, ys_host, ylens_host, blank, True)
            loss = loss_rnnt + eng.mtl_ctc * loss_ctc
        else:
            loss_ctc = torch.zeros_like(loss_rnnt)
        ctx.eng, ctx.st, ctx.cctx, ctx.eouts = eng, st, cctx, eouts
        out_logits = logits.detach().clone() if False else logits
        ctx.mark_non_differentiable(out_logits)
        return loss, loss_rnnt, loss_ctc, out_logits

    @staticmethod
    def backward(ctx, g_total, g_rnnt, g_ctc, _):
        eng = ctx.eng
        deouts = eng.rnnt_backward(ctx.st, (g_total + g_rnnt).to(torch.float32).reshape(1))
        if ctx.cctx is not None:
            g_ctc_eff = (g_total * eng.mtl_ctc + g_ctc).to(torch.float32).reshape(1)
            dcl = eng.ctc_grad(ctx.cctx, 1.0, g_ctc_eff)
            from .. import ops
            deouts = ops.add(deouts, eng.head_backward(ctx.eouts, dcl, "decoder.ctc.output"))
        ctx.st = ctx.cctx = None
        return (None, None, deouts, None, None, None, None, None) + (None,) * len(eng.arena.params)


def rnnt_apply(dec, eouts, elens, ys, ylens, ys_in):
    eng = _engine_of(dec)
    ylens_host = _host_list(ylens)
    L = max(ylens_host)
    ys_host = (ys.cpu() if torch.is_tensor(ys) else torch.as_tensor(ys))[:, :L]
    ys_in = (ys_in.cpu() if torch.is_tensor(ys_in) else torch.as_tensor(ys_in))[:, : L + 1]
    return _RNNTFn.apply(eng, dec.training, eouts, _elens_dev(eouts, elens), ys_host, ylens_host, ys_in, dec.blank_id,
                         *eng.arena.params)


def rnnt_greedy_apply(dec, eouts, elens):
    eng = _engine_of(dec)
    return eng.rnnt_greedy(eouts, _host_list(elens), dec.blank_id, dec.eos_id, dec.max_seq_len)


def rnnt_beam_apply(dec, eouts, beam_width):
    return _engine_of(dec).rnnt_beam_search(eouts, beam_width, dec.blank_id, dec.eos_id)
