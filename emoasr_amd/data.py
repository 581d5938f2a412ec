"""Synthetic LibriSpeech-shaped data, the reference's batch packing rule and SpecAugment span
sampling (host side), as specified in SURVEY.md section 8d.

  ASRBatchSampler rule   asr/datasets.py:204-232  (greedy consecutive packing of the length-sorted
                         list until sum(xlen) > max_xlens_batch, sum(ylen) > max_ylens_batch or
                         len > batch_size; batches smaller than min_batch_size are dropped)
  SpecAugment sampling   asr/spec_augment.py:39-95 (start drawn from randrange(0, dim - f), width is
                         the *second* draw, skipped when f == 0)
"""
import random

import numpy as np


def libri_shaped_lengths(n, seed=0):
    rng = np.random.RandomState(seed)
    xlens = np.clip(np.round(np.exp(rng.normal(np.log(1230.0), 0.45, size=n))), 200, 3500).astype(np.int64)
    xlens.sort()
    ylens = np.maximum(1, np.round(xlens / 30.0)).astype(np.int64)
    return xlens, ylens


def pack_batches(xlens, ylens, max_xlens_batch=30000, max_ylens_batch=3000, batch_size=50, min_batch_size=1):
    batches, cur, sx, sy = [], [], 0, 0
    for i, (x, y) in enumerate(zip(xlens, ylens)):
        if cur and (sx + x > max_xlens_batch or sy + y > max_ylens_batch or len(cur) + 1 > batch_size):
            if len(cur) >= min_batch_size:
                batches.append(cur)
            cur, sx, sy = [], 0, 0
        cur.append(i)
        sx += int(x)
        sy += int(y)
    if len(cur) >= min_batch_size:
        batches.append(cur)
    return batches


ADAPTIVE_MAX_MASKS = 20  # spec_augment.py:72-73: both adaptive quantities are capped at 20


def specaug_spans(xlens, feat_dim, max_mask_freq=30, num_masks_freq=2, max_mask_time=40, num_masks_time=2,
                  np_rng=None, py_rng=None, max_mask_time_ratio=None, num_masks_time_ratio=None):
    """-> int32 array [B, nf+nt, 2] of (start, end) bands; empty bands are (0, 0).

    With `max_mask_time_ratio` / `num_masks_time_ratio` (adaptive SpecAugment, spec_augment.py:20-26,71-73) the width
    bound and the number of time masks follow the utterance: min(20, round(xlen * ratio)) each, so nt = 20 slots are
    returned and the unused ones stay empty.  Like the reference, an utterance so short that the width bound rounds to 0
    raises (numpy's randint(0, 0))."""
    np_rng = np_rng or np.random
    py_rng = py_rng or random
    B = len(xlens)
    adaptive = max_mask_time_ratio is not None
    nt_slots = ADAPTIVE_MAX_MASKS if adaptive else num_masks_time
    spans = np.zeros((B, num_masks_freq + nt_slots, 2), dtype=np.int32)
    for b, xlen in enumerate(xlens):
        xlen = int(xlen)
        fs = np_rng.randint(0, max_mask_freq, size=(num_masks_freq, 2))
        for m, (f, w) in enumerate(fs):
            f_zero = py_rng.randrange(0, feat_dim - f)
            if f == 0:
                continue
            spans[b, m] = (f_zero, min(f_zero + w, feat_dim))
        if adaptive:
            mmt = min(ADAPTIVE_MAX_MASKS, round(xlen * max_mask_time_ratio))
            nmt = min(ADAPTIVE_MAX_MASKS, round(xlen * num_masks_time_ratio))
        else:
            mmt, nmt = max_mask_time, num_masks_time
        ts = np_rng.randint(0, mmt, size=(nmt, 2))
        for m, (t, w) in enumerate(ts):
            if xlen - t <= 0:
                continue
            t_zero = py_rng.randrange(0, xlen - t)
            if t == 0:
                continue
            spans[b, num_masks_freq + m] = (t_zero, min(t_zero + w, xlen))
    return spans


class SpecAugment:
    """asr/spec_augment.py:10-95 for a whole padded batch on the device: the reference's draws per utterance (same numpy /
    random call order, so a seeded run masks the same bands), ONE kernel for the batch instead of a numpy copy per
    utterance in the data loader (asr/datasets.py:37-40,94-95: training phase and params.spec_augment).
    `replace_with_zero=False` fills with the utterance's mean before masking (the reference re-evaluates the mean after
    every band, a difference of O(masked fraction) of the mean)."""

    def __init__(self, params, np_rng=None, py_rng=None):
        self.max_mask_freq, self.num_masks_freq = params.max_mask_freq, params.num_masks_freq
        self.adaptive_specaug = hasattr(params, "max_mask_time_ratio")
        if self.adaptive_specaug:
            self.max_mask_time_ratio, self.num_masks_time_ratio = params.max_mask_time_ratio, params.num_masks_time_ratio
        else:
            self.max_mask_time, self.num_masks_time = params.max_mask_time, params.num_masks_time
        self.replace_with_zero = params.replace_with_zero
        self.np_rng, self.py_rng = np_rng, py_rng

    def spans(self, xlens, feat_dim):
        kw = dict(max_mask_time_ratio=self.max_mask_time_ratio, num_masks_time_ratio=self.num_masks_time_ratio) \
            if self.adaptive_specaug else dict(max_mask_time=self.max_mask_time, num_masks_time=self.num_masks_time)
        return specaug_spans(xlens, feat_dim, self.max_mask_freq, self.num_masks_freq, np_rng=self.np_rng,
                             py_rng=self.py_rng, **kw)

    def __call__(self, xs, xlens):
        """xs: float32 [B, T, F] on the GPU (masked in place and returned); xlens: host lengths."""
        import torch

        from . import ops
        from .engine import h2d_i32
        sp = self.spans([int(v) for v in xlens], xs.shape[-1])
        nt = sp.shape[1] - self.num_masks_freq
        xl = h2d_i32([int(v) for v in xlens], xs.device)
        fill = None
        if not self.replace_with_zero:
            t = torch.arange(xs.shape[1], device=xs.device)[None, :, None] < xl[:, None, None]
            fill = ((xs * t).sum((1, 2)) / (xl.float() * xs.shape[-1])).contiguous()
        return ops.specaug_apply(xs, h2d_i32(sp, xs.device), self.num_masks_freq, nt, xl, fill)
