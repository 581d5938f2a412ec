"""Oracle part 4 (TEST INFRASTRUCTURE ONLY -- never imported by the product path): knowledge distillation.

CPU restatement of asr/criteria.py:49-288 (DistillLoss, CTCAlignDistillLoss, RNNTWordDistillLoss,
RNNTAlignDistillLoss), asr/modeling/decoders/ctc_aligner.py:96-221 (CTCForcedAligner) and of the kd
branches of the CTC / Transformer decoders (decoders/ctc.py:117-127, decoders/transformer.py:117-126).
Pinned to the reference by tests/golden/kd_tiny.npz (tests/test_oracle_kd.py).
"""
import torch

from .model import ctc_loss, linear


def smoothed_onehot(labels, V, eps):
    """criteria.py:5-15: 1-eps on the label, eps/(V-1) on every other class"""
    q = torch.full(labels.shape + (V,), eps / (V - 1))
    q.scatter_(-1, labels.unsqueeze(-1), 1.0 - eps)
    return q


def cross_entropy_rows(logits, q):
    """-sum_v q[v] * log_softmax(logits)[v] per row"""
    return -(torch.log_softmax(logits, -1) * q).sum(-1)


def distill_loss(logits, ys, soft, ylens, soft_label_weight, lsm_prob=0.0, normalize_length=False,
                 normalize_batch=True):
    """DistillLoss.forward (criteria.py:66-100) -> (loss, loss_soft, loss_hard)"""
    B, L, V = logits.shape
    hard = smoothed_onehot(ys[:, :L], V, lsm_prob)
    tot_s = tot_h = 0.0
    for b in range(B):
        n = int(ylens[b])
        ls = cross_entropy_rows(logits[b, :n], soft[b, :n]).sum()
        lh = cross_entropy_rows(logits[b, :n], hard[b, :n]).sum()
        if normalize_length:
            ls, lh = ls / n, lh / n
        tot_s, tot_h = tot_s + ls, tot_h + lh
    if normalize_batch:
        tot_s, tot_h = tot_s / B, tot_h / B
    return soft_label_weight * tot_s + (1 - soft_label_weight) * tot_h, tot_s, tot_h


def frame_to_label_map(align, blank=0, position="all"):
    """criteria.py:170-215.  align: token per frame (already cut to xlen).  A non-blank frame whose token
    differs from the previous frame's opens the next label; `all` maps every frame of the run, the other
    modes one frame of it (first / (first+last)//2 / last).  -> list, -1 where no label is assigned"""
    n = len(align)
    out = [-1] * n
    runs = []  # (first, last) per label, in order
    for t, tok in enumerate(align):
        if tok == blank:
            continue
        if t == 0 or tok != align[t - 1]:
            runs.append([t, t])
        else:
            runs[-1][1] = t
    for k, (a, b) in enumerate(runs):
        if position == "all":
            for t in range(a, b + 1):
                out[t] = k
        else:
            out[{"left": a, "mid": (a + b) // 2, "right": b}[position]] = k
    return out


def ctc_align_distill_loss(logits, ys, soft, aligns, xlens, ylens, blank=0, soft_label_weight=1.0, position="all",
                           lsm_prob=0.0, normalize_length=True, normalize_batch=True):
    """CTCAlignDistillLoss.forward (criteria.py:125-168)"""
    B, T, V = logits.shape
    hard = smoothed_onehot(ys, V, lsm_prob)
    total = 0.0
    for b in range(B):
        n = int(xlens[b])
        lmap = frame_to_label_map([int(v) for v in aligns[b][:n]], blank, position)
        assert max(lmap) == int(ylens[b]) - 1
        frames = [t for t in range(n) if lmap[t] >= 0]
        idx = torch.tensor([lmap[t] for t in frames], dtype=torch.long)
        z = logits[b, frames]
        ls = cross_entropy_rows(z, soft[b, idx]).sum() if soft_label_weight > 0 else 0.0
        lh = cross_entropy_rows(z, hard[b, idx]).sum() if soft_label_weight < 1 else 0.0
        if normalize_length:
            ls, lh = ls / len(frames), lh / len(frames)
        total = total + soft_label_weight * ls + (1 - soft_label_weight) * lh
    return total / B if normalize_batch else total


def rnnt_word_distill_loss(logits, soft, xlens, ylens, normalize_length=True, normalize_batch=True):
    """RNNTWordDistillLoss.forward (criteria.py:227-247): every frame t < xlen and label position u < ylen
    is pulled towards soft[b,u]"""
    B = logits.shape[0]
    total = 0.0
    for b in range(B):
        T, U = int(xlens[b]), int(ylens[b])
        l = cross_entropy_rows(logits[b, :T, :U], soft[b, :U].unsqueeze(0)).sum()
        total = total + (l / (T * U) if normalize_length else l)
    return total / B if normalize_batch else total


def rnnt_align_distill_loss(logits, ys, soft, aligns, xlens, ylens, normalize_length=True, normalize_batch=True):
    """RNNTAlignDistillLoss.forward (criteria.py:259-288).  The reference's inner loop overwrites `loss_u`,
    so only the LAST label position u = ylen-1 (at frame aligns[b][u]) contributes; reproduced as is."""
    B = logits.shape[0]
    total = 0.0
    for b in range(B):
        U = int(ylens[b])
        u = U - 1
        l = cross_entropy_rows(logits[b, int(aligns[b][u]), u], soft[b, u])
        total = total + (l / U if normalize_length else l)
    return total / B if normalize_batch else total


def ctc_forced_align(log_probs, elens, ys, ylens, blank=0):
    """CTCForcedAligner.__call__ (ctc_aligner.py:139-221) -> int64 [B,T], zeros beyond elens.

    post[t,s] = alpha_t[s] (emission at t included) + beta_t[s] (emission at t excluded) over the blank-
    extended label sequence; then a left-to-right pass that, per frame, takes the arg-max of post over the
    states reachable from the previous frame's choice (stay / next / skip-one unless the two labels are
    equal; frame 0: the first blank or the first label).  -1e10 stands for log 0 like the reference."""
    NEG = -1e10
    B, T, V = log_probs.shape
    out = torch.zeros(B, T, dtype=torch.int64)
    for b in range(B):
        n, L = int(elens[b]), int(ylens[b])
        ext = [blank] * (2 * L + 1)
        ext[1::2] = [int(v) for v in ys[b, :L]]
        S = len(ext)
        y = log_probs[b, :n][:, ext].double()  # [n,S]
        skip = [s >= 2 and ext[s] != ext[s - 2] for s in range(S)]

        def step(prev, reverse):
            cur = torch.full((S,), NEG, dtype=torch.float64)
            for s in range(S):
                if reverse:
                    terms = [prev[s]] + ([prev[s + 1]] if s + 1 < S else []) + \
                            ([prev[s + 2]] if s + 2 < S and skip[s + 2] else [])
                else:
                    terms = [prev[s]] + ([prev[s - 1]] if s >= 1 else []) + ([prev[s - 2]] if skip[s] else [])
                cur[s] = torch.logsumexp(torch.stack(terms), 0)
            return cur

        alpha = torch.full((n, S), NEG, dtype=torch.float64)
        a = torch.full((S,), NEG, dtype=torch.float64)
        a[0] = 0.0
        for t in range(n):
            if t == 0:
                pre = torch.full((S,), NEG, dtype=torch.float64)
                pre[0] = 0.0
                if S > 1:
                    pre[1] = 0.0
            else:
                pre = step(a, False)
            a = pre + y[t]
            alpha[t] = a
        beta_pre = torch.full((n, S), NEG, dtype=torch.float64)
        bpost = None
        for t in reversed(range(n)):
            if bpost is None:
                pre = torch.full((S,), NEG, dtype=torch.float64)
                pre[S - 1] = 0.0
                if S > 1:
                    pre[S - 2] = 0.0
            else:
                pre = step(bpost, True)
            beta_pre[t] = pre
            bpost = pre + y[t]
        post = alpha + beta_pre
        cur = None
        for t in range(n):
            if cur is None:
                reach = [0, 1] if S > 1 else [0]
            else:
                reach = [cur] + ([cur + 1] if cur + 1 < S else []) + \
                        ([cur + 2] if cur + 2 < S and skip[cur + 2] else [])
            cur = max(reach, key=lambda s: (post[t, s].item(), -s))
            out[b, t] = ext[cur]
    return out


def ctc_decoder_forward_kd(sd, cfg, eouts, elens, ys, ylens, soft_labels, prefix="decoder"):
    """CTCDecoder.forward with kd_weight > 0 (decoders/ctc.py:103-127,172-174) -> (loss, loss_dict, logits, aligns)"""
    logits = linear(sd, prefix + ".output", eouts)
    loss_ctc = ctc_loss(logits, ys, elens, ylens, cfg.blank_id)
    aligns = ctc_forced_align(torch.log_softmax(logits.detach(), -1), elens, ys, ylens, cfg.blank_id)
    loss_kd = ctc_align_distill_loss(logits, ys, soft_labels, aligns, elens, ylens, blank=cfg.blank_id,
                                     soft_label_weight=getattr(cfg, "kd_ctc_soft_label_weight", 1.0),
                                     position=getattr(cfg, "kd_ctc_position", "all"), lsm_prob=cfg.lsm_prob)
    if cfg.reduce_main_loss_kd:
        loss = (1 - cfg.kd_weight) * loss_ctc + cfg.kd_weight * loss_kd
    else:
        loss = loss_ctc + cfg.kd_weight * loss_kd
    return loss, {"loss_ctc": loss_ctc, "loss_kd": loss_kd, "loss_total": loss}, logits, aligns


def att_decoder_forward_kd(sd, cfg, eouts, elens, ys, ylens, ys_in, ys_out, soft_labels, prefix="decoder"):
    """TransformerDecoder.forward with kd_weight > 0 (decoders/transformer.py:117-146): DistillLoss over
    ys_out with soft_label_weight = kd_weight; the auxiliary CTC gets no distillation."""
    from .decoder import decoder_logits
    logits = decoder_logits(sd, cfg, eouts, elens, ys_in, ylens + 1, prefix)
    loss, loss_kd, loss_att = distill_loss(logits, ys_out, soft_labels, ylens + 1, cfg.kd_weight, cfg.lsm_prob,
                                           cfg.loss_normalize_length, cfg.loss_normalize_batch)
    ld = {"loss_kd": loss_kd, "loss_att": loss_att}
    if cfg.mtl_ctc_weight > 0:
        loss_ctc = ctc_loss(linear(sd, prefix + ".ctc.output", eouts), ys, elens, ylens, cfg.blank_id)
        loss = loss + cfg.mtl_ctc_weight * loss_ctc
        ld["loss_ctc"] = loss_ctc
    ld["loss_total"] = loss
    return loss, ld, logits


def encoder_forward_inter(sd, cfg, xs, xlens, training=False):
    """encoder output plus the intermediate branch: the SAME final LayerNorm applied to the output of layer
    `inter_ctc_layer_id` (encoders/transformer.py:75-82,104-107) -> (eouts, elens, eouts_inter | None)"""
    from .model import encoder_forward, layer_norm
    outs = []
    eouts, elens = encoder_forward(sd, cfg, xs, xlens, training, collect=outs)
    on = getattr(cfg, "mtl_inter_ctc_weight", 0) > 0 or getattr(cfg, "mtl_phone_ctc_weight", 0) > 0
    inter = layer_norm(sd, "encoder.norm", outs[cfg.inter_ctc_layer_id], 1e-12) if on and cfg.inter_ctc_layer_id > 0 else None
    return eouts, elens, inter


def ctc_decoder_forward_full(sd, cfg, eouts, elens, eouts_inter, ys, ylens, soft_labels=None, ps=None, plens=None,
                             prefix="decoder"):
    """CTCDecoder.forward with all its auxiliary branches (decoders/ctc.py:87-174): main CTC, alignment KD,
    phone-level CTC on the final or the intermediate layer, intermediate CTC with optional KD."""
    kd_w = cfg.kd_weight
    kd_kw = dict(blank=cfg.blank_id, soft_label_weight=getattr(cfg, "kd_ctc_soft_label_weight", 1.0),
                 position=getattr(cfg, "kd_ctc_position", "all"), lsm_prob=cfg.lsm_prob)
    logits = linear(sd, prefix + ".output", eouts)
    loss_ctc = ctc_loss(logits, ys, elens, ylens, cfg.blank_id)
    loss, ld = loss_ctc, {"loss_ctc": loss_ctc}
    if kd_w > 0 and soft_labels is not None:
        aligns = ctc_forced_align(torch.log_softmax(logits.detach(), -1), elens, ys, ylens, cfg.blank_id)
        ld["loss_kd"] = ctc_align_distill_loss(logits, ys, soft_labels, aligns, elens, ylens, **kd_kw)
        loss = (1 - kd_w) * loss + kd_w * ld["loss_kd"] if cfg.reduce_main_loss_kd else loss + kd_w * ld["loss_kd"]
    w_ph = getattr(cfg, "mtl_phone_ctc_weight", 0)
    if w_ph > 0:
        src = eouts_inter if cfg.hie_mtl_phone else eouts
        lph = ctc_loss(linear(sd, prefix + ".phone_output", src), ps, elens, plens, cfg.blank_id)
        loss = loss + w_ph * lph
        ld["loss_phone_ctc(inter)" if cfg.hie_mtl_phone else "loss_phone_ctc"] = lph
    w_in = getattr(cfg, "mtl_inter_ctc_weight", 0)
    if w_in > 0:
        logits_i = linear(sd, prefix + ".output", eouts_inter)
        li = ctc_loss(logits_i, ys, elens, ylens, cfg.blank_id)
        ld["loss_inter_ctc"] = li
        w_ikd = getattr(cfg, "inter_kd_weight", 0)
        if w_ikd > 0:
            al = ctc_forced_align(torch.log_softmax(logits_i.detach(), -1), elens, ys, ylens, cfg.blank_id)
            lk = ctc_align_distill_loss(logits_i, ys, soft_labels, al, elens, ylens, **kd_kw)
            ld["loss_inter_kd"] = lk
            if cfg.reduce_main_loss_kd:
                loss = loss + w_in * ((1 - w_ikd) * li + w_ikd * lk)
            else:
                loss = loss + w_ikd * lk  # ctc.py:167 -- the intermediate CTC loss itself is not added here
        else:
            loss = loss + w_in * li
    ld["loss_total"] = loss
    return loss, ld, logits


def rnnt_decoder_forward_kd(sd, cfg, eouts, elens, ys, ylens, ys_in, soft_labels, prefix="decoder"):
    """RNNTDecoder.forward with word-level distillation (rnn_transducer.py:81-145)"""
    from .rnnt import rnnt_decoder_forward
    loss, ld, logits = rnnt_decoder_forward(sd, cfg, eouts, elens, ys, ylens, ys_in, prefix)
    ld = dict(ld)
    ld["loss_kd"] = rnnt_word_distill_loss(logits, soft_labels, elens, ylens)
    if cfg.reduce_main_loss_kd:
        loss = (1 - cfg.kd_weight) * loss + cfg.kd_weight * ld["loss_kd"]
    else:
        loss = loss + cfg.kd_weight * ld["loss_kd"]
    ld["loss_total"] = loss
    return loss, ld, logits


def rnnt_alpha_beta(lp, lab, T, U, blank=0):
    """transducer forward / backward variables of ONE utterance (rnnt_aligner.py:14-152): lp [T', U'+1, V] log-probabilities
    (float64), lab the U labels -> alpha, beta [T, U+1]; alpha[0,0] = 0, beta[T-1,U] = the final blank.
    tests/golden/rnnt_align_xcheck.npz holds what the reference's own recursion bodies gave (make_golden.py runs them in plain
    Python); tests/test_oracle_rnnt.py compares."""
    NEG = float("-inf")
    alpha = torch.full((T, U + 1), NEG, dtype=torch.float64)
    beta = torch.full((T, U + 1), NEG, dtype=torch.float64)
    for t in range(T):
        for u in range(U + 1):
            if t == 0 and u == 0:
                alpha[0, 0] = 0.0
                continue
            a = alpha[t - 1, u] + lp[t - 1, u, blank] if t > 0 else torch.tensor(NEG, dtype=torch.float64)
            e = alpha[t, u - 1] + lp[t, u - 1, lab[u - 1]] if u > 0 else torch.tensor(NEG, dtype=torch.float64)
            alpha[t, u] = torch.logaddexp(a, e)
    for t in reversed(range(T)):
        for u in reversed(range(U + 1)):
            if t == T - 1 and u == U:
                beta[t, u] = lp[t, u, blank]
                continue
            a = beta[t + 1, u] + lp[t, u, blank] if t < T - 1 else torch.tensor(NEG, dtype=torch.float64)
            e = beta[t, u + 1] + lp[t, u, lab[u]] if u < U else torch.tensor(NEG, dtype=torch.float64)
            beta[t, u] = torch.logaddexp(a, e)
    return alpha, beta


def rnnt_forced_align(log_probs, elens, ys, ylens, blank=0):
    """RNNTForcedAligner.__call__ (rnnt_aligner.py:158-198) -> int32 [B, maxU-1].

    alpha[t,u] / beta[t,u]: the transducer forward / backward variables (rnnt_aligner.py:14-152, beta[T-1,U]
    = the final blank); the walk starts at (0,0) and, while t+1 < T and u < U, moves down in time when
    (alpha+beta)[t+1,u] > (alpha+beta)[t,u+1], else emits label u at frame t.  Labels not emitted before the
    last frame keep 0.  PARITY UNPINNED for the lattice part: the reference's Numba CUDA kernels cannot run
    here (numba absent); the recursion is the standard one also used by oracle.rnnt.rnnt_nll.  Cross-check (not a pin: the
    kernels' bodies executed as plain Python, one thread after the other): tests/golden/rnnt_align_xcheck.npz."""
    B, Tm, Um, _ = log_probs.shape
    out = torch.zeros(B, Um - 1, dtype=torch.int32)
    lp = log_probs.double()
    for b in range(B):
        T, U = int(elens[b]), int(ylens[b])
        lab = [int(v) for v in ys[b, :U]]
        alpha, beta = rnnt_alpha_beta(lp[b], lab, T, U, blank)
        post = alpha + beta
        t = u = 0
        while t + 1 < T and u < U:
            if post[t + 1, u] > post[t, u + 1]:
                t += 1
            else:
                out[b, u] = t
                u += 1
    return out


def rnnt_decoder_forward_kd_align(sd, cfg, eouts, elens, ys, ylens, ys_in, soft_labels, prefix="decoder"):
    """RNNTDecoder.forward with kd_type == "align" (rnn_transducer.py:127-141)"""
    from .rnnt import rnnt_decoder_forward
    loss, ld, logits = rnnt_decoder_forward(sd, cfg, eouts, elens, ys, ylens, ys_in, prefix)
    ld = dict(ld)
    aligns = rnnt_forced_align(torch.log_softmax(logits.detach(), -1), elens, ys, ylens, cfg.blank_id)
    ld["loss_kd"] = rnnt_align_distill_loss(logits, ys, soft_labels, aligns, elens, ylens)
    if cfg.reduce_main_loss_kd:
        loss = (1 - cfg.kd_weight) * loss + cfg.kd_weight * ld["loss_kd"]
    else:
        loss = loss + cfg.kd_weight * ld["loss_kd"]
    ld["loss_total"] = loss
    return loss, ld, logits, aligns
