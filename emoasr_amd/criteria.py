"""Loss modules with the signatures of asr/criteria.py.

LabelSmoothingLoss (criteria.py:18-46) runs on the HIP kernel `emoasr_lsm_loss`.  The distillation
losses (criteria.py:49-288) are outside the hot path (every kd weight is 0 in the L-series
configs); their classes exist so that code importing them keeps working, and raise when called.
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


class _OffPath(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, *a, **k):
        raise NotImplementedError(f"emoasr_amd: {type(self).__name__} is outside the HIP hot path (kd weights are 0)")


class DistillLoss(_OffPath):
    pass


class CTCAlignDistillLoss(_OffPath):
    pass


class RNNTWordDistillLoss(_OffPath):
    pass


class RNNTAlignDistillLoss(_OffPath):
    pass
