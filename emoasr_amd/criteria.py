"""Loss modules with the signatures of asr/criteria.py.

LabelSmoothingLoss (criteria.py:18-46) runs on the HIP kernel `emoasr_lsm_loss`; the four distillation
losses (criteria.py:49-288) share the soft-target cross-entropy kernel `emoasr_soft_ce` (csrc/distill.hip).
Index / weight vectors are built with a few tiny device-side tensor ops; there is no host round trip
beyond the length lists the reference also reads.
"""
import torch
import torch.nn as nn

from . import ops


class _LsmFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, w, lsm):
        B, L, V = logits.shape
        rows, _ = ops.lsm_loss(logits.reshape(B * L, V), labels.view(-1), w.view(-1), lsm)
        ctx.save_for_backward(logits, labels, w)
        ctx.lsm = lsm
        return rows.sum()

    @staticmethod
    def backward(ctx, g):
        logits, labels, w = ctx.saved_tensors
        B, L, V = logits.shape
        _, grad = ops.lsm_loss(logits.reshape(B * L, V), labels.view(-1), w.view(-1), ctx.lsm, True, 1.0,
                               g.to(torch.float32).reshape(1))
        return grad.view(B, L, V), None, None, None


class LabelSmoothingLoss(nn.Module):
    def __init__(self, vocab_size, lsm_prob=0, normalize_length=False, normalize_batch=True):
        super().__init__()
        self.vocab_size = vocab_size
        self.lsm_prob = lsm_prob
        self.normalize_length = normalize_length
        self.normalize_batch = normalize_batch

    def forward(self, logits, ys, ylens):
        """logits [B,L,V] (device), ys [B,L] int64, ylens [B] -> 0-dim loss"""
        B, L, V = logits.shape
        w = torch.zeros(B, L, dtype=torch.float32)
        for b in range(B):
            n = int(ylens[b])
            w[b, :n] = (1.0 / B if self.normalize_batch else 1.0) / (n if self.normalize_length else 1.0)
        dev = logits.device
        labels = torch.as_tensor(ys)[:, :L].to(torch.int32).contiguous().to(dev)
        return _LsmFn.apply(logits.contiguous(), labels, w.to(dev), float(self.lsm_prob))


class _SoftCEFn(torch.autograd.Function):
    """sum over rows of the soft/hard cross-entropy kernel (`emoasr_soft_ce`); the two weight vectors are
    scaled on the device by the incoming gradients, so one kernel pass produces the logits gradient."""

    @staticmethod
    def forward(ctx, logits2d, soft2d, src, hard, w_soft, w_hard, lsm, lrow):
        rs, _ = ops.soft_ce(logits2d, soft2d, src, None, w_soft, None, lsm, lrow)
        rh, _ = ops.soft_ce(logits2d, None, None, hard, None, w_hard, lsm, lrow) if w_hard is not None else (None, None)
        ctx.save_for_backward(logits2d, soft2d, src, hard, w_soft, w_hard, lrow)
        ctx.lsm = lsm
        ls = rs.sum()
        return ls, (rh.sum() if rh is not None else torch.zeros_like(ls))

    @staticmethod
    def backward(ctx, gs, gh):
        logits2d, soft2d, src, hard, w_soft, w_hard, lrow = ctx.saved_tensors
        ws = w_soft * gs.to(torch.float32)
        wh = w_hard * gh.to(torch.float32) if w_hard is not None else None
        _, grad = ops.soft_ce(logits2d, soft2d, src, hard if wh is not None else None, ws, wh, ctx.lsm, lrow,
                              want_grad=True)
        return grad, None, None, None, None, None, None, None


def _i32(t, dev):
    return torch.as_tensor(t).to(device=dev, dtype=torch.int32)


def _soft(soft_labels, dev):
    return torch.as_tensor(soft_labels).to(device=dev, dtype=torch.float32).contiguous()


class DistillLoss(nn.Module):
    """criteria.py:49-100: label interpolation between the teacher's soft labels and the label-smoothed
    hard labels -> (loss, loss_soft, loss_hard)"""

    def __init__(self, vocab_size, soft_label_weight, lsm_prob=0, normalize_length=False, normalize_batch=True):
        super().__init__()
        self.vocab_size = vocab_size
        self.soft_label_weight = soft_label_weight
        self.lsm_prob = lsm_prob
        self.normalize_length = normalize_length
        self.normalize_batch = normalize_batch

    def forward(self, logits, ys, soft_labels, ylens):
        B, L, V = logits.shape
        dev = logits.device
        w = torch.zeros(B, L, dtype=torch.float32)
        for b in range(B):
            n = int(ylens[b])
            w[b, :n] = (1.0 / B if self.normalize_batch else 1.0) / (n if self.normalize_length else 1.0)
        w = w.to(dev).view(-1)
        soft = _soft(soft_labels, dev)[:, :L]
        assert soft.shape == (B, L, V), (tuple(soft.shape), (B, L, V))
        src = torch.arange(B * L, device=dev, dtype=torch.int32)
        hard = _i32(ys, dev)[:, :L].contiguous().view(-1)
        ls, lh = _SoftCEFn.apply(logits.contiguous().view(B * L, V), soft.contiguous().view(B * L, V), src, hard, w, w,
                                 float(self.lsm_prob), None)
        return self.soft_label_weight * ls + (1 - self.soft_label_weight) * lh, ls, lh


class CTCAlignDistillLoss(nn.Module):
    """criteria.py:103-215: every frame that the forced alignment assigns to a label is pulled towards that
    label's soft target (and/or its smoothed one-hot); the frame -> label map is the HIP kernel
    `emoasr_ctc_label_map`, the loss `emoasr_soft_ce` with the soft rows gathered by index."""

    def __init__(self, vocab_size, blank_id=0, soft_label_weight=1.0, position="all", lsm_prob=0,
                 normalize_length=True, normalize_batch=True):
        super().__init__()
        self.vocab_size = vocab_size
        self.blank_id = blank_id
        self.soft_label_weight = soft_label_weight
        self.position = position
        self.lsm_prob = lsm_prob
        self.normalize_length = normalize_length
        self.normalize_batch = normalize_batch

    def forward(self, logits, ys, soft_labels, aligns, xlens, ylens):
        B, T, V = logits.shape
        dev = logits.device
        al = _i32(aligns, dev)[:, :T].contiguous()
        if al.shape[1] < T:
            al = torch.nn.functional.pad(al, (0, T - al.shape[1]))
        lmap, count = ops.ctc_label_map(al, _i32(xlens, dev), self.blank_id, self.position)
        soft = _soft(soft_labels, dev)
        L = soft.shape[1]
        ysd = _i32(ys, dev)
        has = lmap >= 0
        idx = lmap.clamp(min=0)
        src = torch.where(has, idx + torch.arange(B, device=dev, dtype=torch.int32)[:, None] * L, -1).to(torch.int32)
        hard = torch.where(has, ysd.gather(1, idx.long().clamp(max=ysd.shape[1] - 1)), -1).to(torch.int32)
        w = torch.full((B, 1), 1.0 / B if self.normalize_batch else 1.0, device=dev)
        if self.normalize_length:
            w = w / count.clamp(min=1).to(torch.float32)[:, None]
        w = w.expand(B, T).contiguous().view(-1)
        a = float(self.soft_label_weight)
        ls, lh = _SoftCEFn.apply(logits.contiguous().view(B * T, V), soft.view(B * L, V), src.view(-1), hard.view(-1),
                                 w * a, (w * (1 - a)) if a < 1 else None, float(self.lsm_prob), None)
        return ls + lh


def rnnt_word_rows(B, T, U, L, xlens, ylens, dev, normalize_length=True, normalize_batch=True):
    """row vectors of the word-level transducer distillation (criteria.py:227-247): for the cell (b,t,u) the
    soft-label row b*L+u (or -1 outside t < xlen, u < ylen) and the weight 1/(xlen*ylen)/B"""
    xl, yl = _i32(xlens, dev), _i32(ylens, dev)
    t = torch.arange(T, device=dev, dtype=torch.int32)[None, :, None]
    u = torch.arange(U, device=dev, dtype=torch.int32)[None, None, :]
    live = (t < xl[:, None, None]) & (u < yl[:, None, None]) & (u < L)
    src = torch.where(live, u + torch.arange(B, device=dev, dtype=torch.int32)[:, None, None] * L, -1).to(torch.int32)
    w = torch.full((B,), 1.0 / B if normalize_batch else 1.0, device=dev)
    if normalize_length:
        w = w / (xl * yl).clamp(min=1).to(torch.float32)
    return src.contiguous().view(-1), w[:, None, None].expand(B, T, U).contiguous().view(-1)


class RNNTWordDistillLoss(nn.Module):
    """criteria.py:218-247: logits [B,T,L+1,V]; every (t < xlen, u < ylen) cell is pulled towards soft[b,u]"""

    def __init__(self, normalize_length=True, normalize_batch=True):
        super().__init__()
        self.normalize_length = normalize_length
        self.normalize_batch = normalize_batch

    def forward(self, logits, soft_labels, xlens, ylens):
        B, T, U, V = logits.shape
        soft = _soft(soft_labels, logits.device)
        L = soft.shape[1]
        src, w = rnnt_word_rows(B, T, U, L, xlens, ylens, logits.device, self.normalize_length, self.normalize_batch)
        ls, _ = _SoftCEFn.apply(logits.contiguous().view(B * T * U, V), soft.view(B * L, V), src, None, w, None, 0.0, None)
        return ls


class RNNTAlignDistillLoss(nn.Module):
    """criteria.py:250-288.  The reference's loop over label positions overwrites its accumulator, so only
    the LAST position u = ylen-1 (cell [aligns[b][u], u]) contributes, divided by ylen; reproduced as is."""

    def __init__(self, normalize_length=True, normalize_batch=True):
        super().__init__()
        self.normalize_length = normalize_length
        self.normalize_batch = normalize_batch

    def forward(self, logits, ys, soft_labels, aligns, xlens, ylens):
        B, T, U, V = logits.shape
        dev = logits.device
        soft = _soft(soft_labels, dev)
        L = soft.shape[1]
        yl = _i32(ylens, dev)
        u = (yl - 1).clamp(min=0)
        al = _i32(aligns, dev)
        t = al.gather(1, u.long()[:, None].clamp(max=al.shape[1] - 1))[:, 0]
        b = torch.arange(B, device=dev, dtype=torch.int32)
        lrow = ((b * T + t) * U + u).to(torch.int32)
        src = (b * L + u).to(torch.int32)
        w = torch.full((B,), 1.0 / B if self.normalize_batch else 1.0, device=dev)
        if self.normalize_length:
            w = w / yl.clamp(min=1).to(torch.float32)
        ls, _ = _SoftCEFn.apply(logits.contiguous().view(B * T * U, V), soft.view(B * L, V), src, None, w, None, 0.0, lrow)
        return ls
