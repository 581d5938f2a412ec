"""autograd glue: two Functions connect the engine's explicit forward/backward to
`loss.backward()`, one for the encoder stack and one for the CTC head, so the
encoder/decoder module boundary of the reference survives.

Parameter gradients are accumulated by the engine directly into the flat gradient arena
(the parameters' .grad tensors are views of it); the Functions therefore return None for
the parameter inputs.
"""
import torch

from ..engine import h2d_i32


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
        inter = eng.eouts_inter
        ctx.has_inter = inter is not None
        if inter is None:
            inter = eouts.new_empty(0)
            ctx.mark_non_differentiable(inter)
        return eouts, elens_dev, inter

    @staticmethod
    def backward(ctx, deouts, _, dinter):
        ctx.eng.backward(ctx.st, deouts.contiguous(), dinter.contiguous() if ctx.has_inter else None)
        ctx.st = None
        return (None,) * (4 + len(ctx.eng.arena.params))


def encoder_apply(enc, xs, xlens):
    eng = _engine_of(enc)
    xs = xs.to(torch.float32).contiguous()
    host = _host_list(xlens)
    if torch.is_grad_enabled():
        eng.step_count += 1
        eouts, elens_dev, inter = _EncoderFn.apply(eng, enc.training, xs, host, *eng.arena.params)
        inter = inter if eng.eouts_inter is not None else None
    else:
        eouts, _, elens_dev, _ = eng.forward(xs, host, enc.training, stash=False)
        inter = eng.eouts_inter
    elens = torch.tensor([((v - 1) // 2 - 1) // 2 for v in host], dtype=torch.int64)
    if torch.is_tensor(xlens):
        elens = elens.to(xlens.device)
    eouts._emo_elens_dev = elens_dev
    if inter is not None:
        inter._emo_elens_dev = elens_dev
    return eouts, elens, inter


class _EncoderStackedFn(torch.autograd.Function):
    """the encoder over several micro-batches in ONE stacked pass (engine.encoder_forward_stacked): outputs one eouts tensor per
    micro-batch (views of the stacked rows); the backward waits for all of their gradients and runs one stacked sweep"""

    @staticmethod
    def forward(ctx, eng, n, *args):
        xs_list, xlens_list = list(args[:n]), list(args[n:2 * n])
        eouts, st = eng.encoder_forward_stacked(xs_list, xlens_list)
        ctx.eng, ctx.st = eng, st
        outs = tuple(eouts[st.rows[k]:st.rows[k + 1]].view(st.segs[k][0], st.segs[k][1], eouts.shape[1]) for k in range(n))
        return outs

    @staticmethod
    def backward(ctx, *grads):
        st = ctx.st
        eng = ctx.eng
        d = grads[0].shape[-1] if grads[0] is not None else eng.d
        deouts = torch.empty(st.M, d, device=st.y2.device, dtype=eng.dtype)
        for k, gk in enumerate(grads):
            dst = deouts[st.rows[k]:st.rows[k + 1]]
            if gk is None:
                dst.zero_()
            else:
                dst.copy_(gk.reshape(-1, d))
        eng.encoder_backward_stacked(st, deouts)
        ctx.st = None
        return (None, None) + (None,) * (len(grads) * 2 + len(eng.arena.params))


def encoder_apply_stacked(enc, xs_list, xlens_list):
    """TransformerEncoder.forward for several micro-batches at once (training, autograd on): -> [(eouts, elens, None), ...].
    Each micro-batch keeps its own padding, BatchNorm batch statistics and relative-position table (include/emoasr_hip.h:
    emoasr_segments_t); only the row-wise kernels see them together."""
    eng = _engine_of(enc)
    xs_list = [x.to(torch.float32).contiguous() for x in xs_list]
    hosts = [_host_list(v) for v in xlens_list]
    n = len(xs_list)
    outs = _EncoderStackedFn.apply(eng, n, *xs_list, *hosts, *eng.arena.params)
    res = []
    for k in range(n):
        elens = torch.tensor([((v - 1) // 2 - 1) // 2 for v in hosts[k]], dtype=torch.int64)
        eo = outs[k]
        # the decoders' device copy of the lengths, uploaded from pinned memory without waiting for the stream (a pageable
        # `.to(device)` here blocked the host behind the whole stacked encoder pass, once per micro-batch: the launch queue ran
        # dry five times per optimizer step)
        eo._emo_elens_dev = h2d_i32(elens.to(torch.int32), eo.device)
        res.append((eo, elens, None))
    return res


class _CTCLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, eouts, elens_dev, ys_host, ylens_host, blank, head, *params):
        logits = eng.head_logits(eouts, head)
        need = eouts.requires_grad or any(p.requires_grad for p in params)
        loss, cctx = eng.ctc_loss(logits, elens_dev, ys_host, ylens_host, blank, need)
        ctx.eng, ctx.cctx, ctx.eouts, ctx.head = eng, cctx, eouts, head
        ctx.mark_non_differentiable(logits)
        return loss, logits

    @staticmethod
    def backward(ctx, gloss, _):
        eng = ctx.eng
        gdev = gloss.to(torch.float32).reshape(1) if gloss.is_cuda else None  # no host sync
        dlogits = eng.ctc_grad(ctx.cctx, 1.0 if gdev is not None else float(gloss), gdev)
        deouts = eng.head_backward(ctx.eouts, dlogits, ctx.head)
        ctx.cctx = None
        return (None, deouts, None, None, None, None, None) + (None,) * len(eng.arena.params)


class _HeadFn(torch.autograd.Function):
    """a vocabulary head on its own (logits = eouts W^T + b), for losses that are separate autograd nodes"""

    @staticmethod
    def forward(ctx, eng, eouts, head, *params):
        ctx.eng, ctx.eouts, ctx.head = eng, eouts, head
        return eng.head_logits(eouts, head)

    @staticmethod
    def backward(ctx, dlogits):
        deouts = ctx.eng.head_backward(ctx.eouts, dlogits, ctx.head)
        return (None, deouts, None) + (None,) * len(ctx.eng.arena.params)


class _CTCLogitsFn(torch.autograd.Function):
    """CTC loss of given logits (+ optionally the forced alignment read off the same lattices)"""

    @staticmethod
    def forward(ctx, eng, logits, elens_dev, ys_host, ylens_host, blank, want_aligns):
        from .. import ops
        loss, cctx = eng.ctc_loss(logits, elens_dev, ys_host, ylens_host, blank, True)
        ctx.eng, ctx.cctx = eng, cctx
        if want_aligns:
            _, _, labels, elens, ylens, _, lp, alpha, beta, _ = cctx
            aligns = ops.ctc_best_path(lp, alpha, beta, labels, elens, ylens, blank)
        else:
            aligns = torch.empty(0, dtype=torch.int32, device=logits.device)
        ctx.mark_non_differentiable(aligns)
        return loss, aligns

    @staticmethod
    def backward(ctx, gloss, _):
        dlogits = ctx.eng.ctc_grad(ctx.cctx, 1.0, gloss.to(torch.float32).reshape(1))
        ctx.cctx = None
        return None, dlogits, None, None, None, None, None


def head_apply(dec, eouts, head):
    eng = _engine_of(dec)
    return _HeadFn.apply(eng, eouts, head, *eng.arena.params)


def ctc_from_logits_apply(dec, logits, eouts, elens, ys, ylens, want_aligns=False):
    """-> (loss_ctc = sum_b nll_b / B, aligns int32 [B,T] | None)"""
    eng = _engine_of(dec)
    ys_host = ys.cpu() if torch.is_tensor(ys) else torch.as_tensor(ys)
    loss, aligns = _CTCLogitsFn.apply(eng, logits, _elens_dev(eouts, elens), ys_host, _host_list(ylens), dec.blank_id,
                                      want_aligns)
    return loss, (aligns if want_aligns else None)


def ctc_forced_align_apply(log_probs, elens, ys, ylens, blank):
    """CTCForcedAligner.__call__ (ctc_aligner.py:139-221) on the CTC lattice kernels -> int64 [B,T] (device)"""
    from .. import ops
    dev = log_probs.device
    lp_in = log_probs.contiguous()
    B, T, V = lp_in.shape
    ylens_host = _host_list(ylens)
    Lmax = max(max(ylens_host), 1)
    labels = torch.as_tensor(ys)[:, :Lmax].to(torch.int32)
    if labels.shape[1] < Lmax:
        labels = torch.nn.functional.pad(labels, (0, Lmax - labels.shape[1]))
    labels = labels.contiguous().to(dev)
    el = torch.as_tensor(elens).to(torch.int32).to(dev)
    yl = torch.tensor(ylens_host, dtype=torch.int32).to(dev)
    with ops.stream_scope():
        lse = ops.row_lse(lp_in.view(B * T, V))
        lp, alpha, beta, _ = ops.ctc_forward(lp_in, lse, labels, el, yl, blank)
        return ops.ctc_best_path(lp, alpha, beta, labels, el, yl, blank).long()


def _elens_dev(eouts, elens):
    dev = getattr(eouts, "_emo_elens_dev", None)
    if dev is None:
        t = torch.as_tensor(elens)
        # (a host tensor goes up from pinned memory, asynchronously; a device tensor is converted in place on the stream)
        dev = t.to(torch.int32) if t.is_cuda else h2d_i32(t.to(torch.int32), eouts.device)
    return dev


def ctc_head_apply(dec, eouts, head="decoder.output"):
    return _engine_of(dec).head_logits(eouts, head)


def ctc_loss_apply(dec, eouts, elens, ys, ylens, head="decoder.output"):
    eng = _engine_of(dec)
    ys_host = ys.cpu() if torch.is_tensor(ys) else torch.as_tensor(ys)
    return _CTCLossFn.apply(eng, eouts, _elens_dev(eouts, elens), ys_host, _host_list(ylens), dec.blank_id, head,
                            *eng.arena.params)


def ctc_greedy_apply(dec, eouts, elens):
    eng = _engine_of(dec)
    logits = eng.head_logits(eouts, getattr(dec, "_prefix", "decoder") + ".output", out_f32=eng.f32_head)
    best, hyp, hyplen = eng.greedy(logits, _elens_dev(eouts, elens), dec.blank_id)
    B, T = best.shape
    packed = best._base.cpu() if best._base is not None and best._base.numel() == 2 * B * T + B else None   # one D2H per batch
    if packed is not None:
        best_h, hyp_h, n_h = packed[:B * T].view(B, T), packed[B * T:2 * B * T].view(B, T), packed[2 * B * T:].tolist()
    else:
        best_h, hyp_h, n_h = best.cpu(), hyp.cpu(), hyplen.cpu().tolist()
    el = _host_list(elens)
    hyps = [hyp_h[b, : n_h[b]].tolist() for b in range(len(n_h))]
    aligns = [best_h[b, : el[b]].tolist() for b in range(len(n_h))]
    return hyps, [None] * len(hyps), logits, aligns


# ---------------------------------------------------------------------------------------
# Transformer decoder: attention loss (+ auxiliary CTC) as one autograd node over the encoder output
# ---------------------------------------------------------------------------------------
class _AttnDecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, training, eouts, elens_dev, ys_host, ylens_host, ys_in, ys_out, blank, soft, kd, *params):
        logits, st = eng.dec_forward(eouts, elens_dev, ys_in, ylens_host, training, True)
        if soft is None:
            loss_att, _ = eng.att_loss(logits, ys_out, ylens_host)
            loss_kd, loss = torch.zeros_like(loss_att), loss_att
        else:  # label interpolation with the teacher's soft labels (decoders/transformer.py:117-126)
            loss_kd, loss_att, _ = eng.att_kd_loss(logits, ys_out, ylens_host, soft)
            loss = kd * loss_kd + (1 - kd) * loss_att
        loss_ctc, cctx = None, None
        ctx.soft, ctx.kd = soft, kd
        if eng.mtl_ctc > 0:
            ctc_logits = eng.head_logits(eouts, "decoder.ctc.output")
            loss_ctc, cctx = eng.ctc_loss(ctc_logits, elens_dev, ys_host, ylens_host, blank, True)
            loss = loss + eng.mtl_ctc * loss_ctc
        else:
            loss_ctc = torch.zeros_like(loss_att)
        ctx.eng, ctx.st, ctx.cctx, ctx.eouts, ctx.logits = eng, st, cctx, eouts, logits
        ctx.ys_out, ctx.ylens_host = ys_out, ylens_host
        ctx.mark_non_differentiable(logits)
        return loss, loss_att, loss_ctc, logits, loss_kd

    @staticmethod
    def backward(ctx, g_total, g_att, g_ctc, _, g_kd):
        eng = ctx.eng
        if ctx.soft is None:
            g_att_eff = (g_total + g_att).to(torch.float32).reshape(1)
            _, dlogits = eng.att_loss(ctx.logits, ctx.ys_out, ctx.ylens_host, True, g_att_eff)
        else:
            _, _, dlogits = eng.att_kd_loss(ctx.logits, ctx.ys_out, ctx.ylens_host, ctx.soft,
                                            (g_total * ctx.kd + g_kd).to(torch.float32),
                                            (g_total * (1 - ctx.kd) + g_att).to(torch.float32))
        deouts = eng.dec_backward(ctx.st, dlogits)
        if ctx.cctx is not None:
            g_ctc_eff = (g_total * eng.mtl_ctc + g_ctc).to(torch.float32).reshape(1)
            dcl = eng.ctc_grad(ctx.cctx, 1.0, g_ctc_eff)
            from .. import ops
            deouts = ops.add(deouts, eng.head_backward(ctx.eouts, dcl, "decoder.ctc.output"))
        ctx.st = ctx.cctx = ctx.soft = None
        return (None, None, deouts, None, None, None, None, None, None, None, None) + (None,) * len(eng.arena.params)


def attn_decoder_apply(dec, eouts, elens, ys, ylens, ys_in, ys_out, soft_labels=None, kd_weight=0.0):
    eng = _engine_of(dec)
    ylens_host = _host_list(ylens)
    L = max(ylens_host) + 1
    ys_host = ys.cpu() if torch.is_tensor(ys) else torch.as_tensor(ys)
    ys_in = (ys_in.cpu() if torch.is_tensor(ys_in) else torch.as_tensor(ys_in))[:, :L]
    ys_out = (ys_out.cpu() if torch.is_tensor(ys_out) else torch.as_tensor(ys_out))[:, :L]
    soft = None
    if soft_labels is not None:
        soft = torch.as_tensor(soft_labels)[:, :L].to(device=eouts.device, dtype=torch.float32).contiguous()
    return _AttnDecoderFn.apply(eng, dec.training, eouts, _elens_dev(eouts, elens), ys_host, ylens_host, ys_in, ys_out,
                                dec.blank_id, soft, float(kd_weight), *eng.arena.params)


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
class _PredNetFn(torch.autograd.Function):
    """the RNN-T prediction network of several micro-batches in one pass (engine.rnnt_prediction_stacked)"""

    @staticmethod
    def forward(ctx, eng, training, ys_in_list, *params):
        douts, rst, spans = eng.rnnt_prediction_stacked(ys_in_list, training)
        ctx.eng, ctx.rst = eng, rst
        ctx.spans = spans
        return douts

    @staticmethod
    def backward(ctx, g):
        from .. import ops
        eng = ctx.eng
        with eng._scope():   # (the engine's own product mode: this runs on an autograd thread)
            eng.arena.attach_grads()
            eng.rnnt_recurrency_bwd(ctx.rst, g.contiguous())
        ctx.rst = None
        return (None, None, None) + (None,) * len(eng.arena.params)


def rnnt_prediction_stacked(dec, ys_in_list, ylens_list):
    """-> one [U_k, B_k, H] prediction-network output per micro-batch (inputs of rnnt_apply(..., pred=...)), all computed in ONE
    pass, or None when the stacked pass does not apply (then every micro-batch runs its own, as before)"""
    eng = _engine_of(dec)
    mats = []
    for ys_in, ylens in zip(ys_in_list, ylens_list):
        L = max(_host_list(ylens))
        mats.append((ys_in.cpu() if torch.is_tensor(ys_in) else torch.as_tensor(ys_in))[:, : L + 1])
    if len(mats) < 2 or not eng.rnnt_prediction_stacked_ok(sum(m.shape[0] for m in mats)):
        return None
    douts = _PredNetFn.apply(eng, dec.training, mats, *eng.arena.params)
    out, b0 = [], 0
    for m in mats:
        out.append(douts[: m.shape[1], b0:b0 + m.shape[0]].contiguous())
        b0 += m.shape[0]
    return out


class _RNNTFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, training, eouts, elens_dev, ys_host, ylens_host, ys_in, blank, kd, want_logits, pred, *params):
        """kd = None | (soft f32 [B,L,V] on the device, kd_weight, reduce_main_loss_kd): word-level distillation
        (rnn_transducer.py:127-141, criteria.py:218-247) of every lattice cell towards its label's soft target"""
        from .. import ops
        from ..criteria import rnnt_word_rows
        # the 4-D logits are formed only when something reads them: distillation, or a caller that asked for them
        # (RNNTDecoder.return_logits); training without either runs the fused output layer and returns logits = None
        # with an auxiliary CTC branch the transducer lattice (2 B blocks, ~190 us of a mostly idle chip) runs on a side stream under it
        loss_rnnt, logits, st = eng.rnnt_forward(eouts, elens_dev, ys_in, ys_host, ylens_host, blank, training,
                                                 want_logits=want_logits or kd is not None,
                                                 pred=None if pred is None else pred.detach(),
                                                 defer_lattice=eng.mtl_ctc > 0)
        cctx, loss = None, loss_rnnt
        if eng.mtl_ctc > 0:
            ctc_logits = eng.head_logits(eouts, "decoder.ctc.output")
            loss_ctc, cctx = eng.ctc_loss(ctc_logits, elens_dev, ys_host, ylens_host, blank, True)
            if loss_rnnt is None:
                loss_rnnt = eng.rnnt_lattice_join(st)
            loss = loss_rnnt + eng.mtl_ctc * loss_ctc
        else:
            loss_ctc = torch.zeros_like(loss_rnnt)
        loss_kd = torch.zeros_like(loss_rnnt)
        ctx.kd = None
        if kd is not None:
            soft, kd_w, reduce, kd_type = kd
            B, T, U, V = logits.shape
            L = soft.shape[1]
            if kd_type == "word":
                src, w = rnnt_word_rows(B, T, U, L, elens_dev, ylens_host, logits.device)
                z, lrow = logits.view(B * T * U, V), None
            else:
                # "align" (rnn_transducer.py:131-135): forced alignment from the loss lattices, then
                # RNNTAlignDistillLoss -- whose loop keeps only the LAST label of each utterance
                # (criteria.py:271-281), i.e. one lattice cell per utterance
                alpha, beta = st.ctx[3], st.ctx[4]
                aligns = ops.rnnt_best_path(alpha, beta, st.elens, st.ylens)
                bi = torch.arange(B, device=logits.device, dtype=torch.int32)
                u = (st.ylens - 1).clamp(min=0)
                t = aligns.gather(1, u.long()[:, None].clamp(max=max(aligns.shape[1] - 1, 0)))[:, 0]
                lrow = ((bi * T + t) * U + u).long()
                z = logits.view(B * T * U, V).index_select(0, lrow)  # [B,V], kept: the logits get overwritten
                src = (bi * L + u).to(torch.int32)
                w = (1.0 / B) / st.ylens.clamp(min=1).to(torch.float32)
            rows, _ = ops.soft_ce(z, soft.view(B * L, V), src, None, w, None, 0.0)
            loss_kd = rows.sum()
            loss = ((1 - kd_w) * loss if reduce else loss) + kd_w * loss_kd
            ctx.kd = (soft, kd_w, (1 - kd_w) if reduce else 1.0, src, w, z, lrow)
        ctx.eng, ctx.st, ctx.cctx, ctx.eouts = eng, st, cctx, eouts
        if logits is None:
            logits = eouts.new_empty(0)   # (autograd outputs must be tensors; rnnt_apply turns it back into None)
        ctx.mark_non_differentiable(logits)
        return loss, loss_rnnt, loss_ctc, logits, loss_kd

    @staticmethod
    def backward(ctx, g_total, g_rnnt, g_ctc, _, g_kd):
        from .. import ops
        eng = ctx.eng
        main, extra = 1.0, None
        if ctx.kd is not None:  # before rnnt_backward overwrites the logits with their gradient
            soft, kd_w, main, src, w, z, lrow = ctx.kd
            V = z.shape[-1]
            _, extra = ops.soft_ce(z, soft.view(-1, V), src, None, w * (g_total * kd_w + g_kd).to(torch.float32), None,
                                   0.0, want_grad=True)
            extra = extra if lrow is None else (lrow, extra)
        deouts = eng.rnnt_backward(ctx.st, (g_total * main + g_rnnt).to(torch.float32).reshape(1), extra)
        if ctx.cctx is not None:
            g_ctc_eff = (g_total * (main * eng.mtl_ctc) + g_ctc).to(torch.float32).reshape(1)
            dcl = eng.ctc_grad(ctx.cctx, 1.0, g_ctc_eff)
            deouts = ops.add(deouts, eng.head_backward(ctx.eouts, dcl, "decoder.ctc.output"))
        dpred = getattr(ctx.st, "ddouts", None)   # (the prediction network ran stacked: its gradient goes back to that pass)
        ctx.st = ctx.cctx = ctx.kd = None
        return (None, None, deouts, None, None, None, None, None, None, None, dpred) + (None,) * len(eng.arena.params)


def rnnt_apply(dec, eouts, elens, ys, ylens, ys_in, kd=None, pred=None):
    eng = _engine_of(dec)
    ylens_host = _host_list(ylens)
    L = max(ylens_host)
    ys_host = (ys.cpu() if torch.is_tensor(ys) else torch.as_tensor(ys))[:, :L]
    ys_in = (ys_in.cpu() if torch.is_tensor(ys_in) else torch.as_tensor(ys_in))[:, : L + 1]
    if kd is not None:
        kd = (torch.as_tensor(kd[0]).to(device=eouts.device, dtype=torch.float32).contiguous(), float(kd[1]), bool(kd[2]),
              kd[3])
    want_logits = bool(getattr(dec, "return_logits", False)) or not dec.training
    out = _RNNTFn.apply(eng, dec.training, eouts, _elens_dev(eouts, elens), ys_host, ylens_host, ys_in, dec.blank_id, kd,
                        want_logits, pred, *eng.arena.params)
    loss, loss_rnnt, loss_ctc, logits, loss_kd = out
    return loss, loss_rnnt, loss_ctc, (logits if logits.numel() else None), loss_kd


def rnnt_forced_align_apply(log_probs, elens, ys, ylens, blank):
    """RNNTForcedAligner.__call__ (rnnt_aligner.py:158-198) -> int32 [B,L] (device)"""
    from .. import ops
    dev = log_probs.device
    z = log_probs.contiguous()
    B, T, U, V = z.shape
    labels = torch.as_tensor(ys)[:, : max(U - 1, 1)].to(torch.int32)
    if labels.shape[1] < max(U - 1, 1):
        labels = torch.nn.functional.pad(labels, (0, max(U - 1, 1) - labels.shape[1]))
    labels = labels.contiguous().to(dev)
    el = torch.as_tensor(elens).to(torch.int32).to(dev)
    yl = torch.as_tensor(ylens).to(torch.int32).to(dev)
    with ops.stream_scope():
        (_, _, _, alpha, beta), _ = ops.rnnt_forward(z, labels, el, yl, blank)
        return ops.rnnt_best_path(alpha, beta, el, yl)


def rnnt_greedy_apply(dec, eouts, elens):
    eng = _engine_of(dec)
    return eng.rnnt_greedy(eouts, _host_list(elens), dec.blank_id, dec.eos_id, dec.max_seq_len)


def rnnt_beam_apply(dec, eouts, beam_width):
    return _engine_of(dec).rnnt_beam_search(eouts, beam_width, dec.blank_id, dec.eos_id)
