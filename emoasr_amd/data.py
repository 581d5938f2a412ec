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


def specaug_spans(xlens, feat_dim, max_mask_freq=30, num_masks_freq=2, max_mask_time=40, num_masks_time=2,
                  np_rng=None, py_rng=None):
    """-> int32 array [B, nf+nt, 2] of (start, end) bands; empty bands are (0, 0)."""
    np_rng = np_rng or np.random
    py_rng = py_rng or random
    B = len(xlens)
    spans = np.zeros((B, num_masks_freq + num_masks_time, 2), dtype=np.int32)
    for b, xlen in enumerate(xlens):
        fs = np_rng.randint(0, max_mask_freq, size=(num_masks_freq, 2))
        for m, (f, w) in enumerate(fs):
            f_zero = py_rng.randrange(0, feat_dim - f)
            if f == 0:
                continue
            spans[b, m] = (f_zero, min(f_zero + w, feat_dim))
        ts = np_rng.randint(0, max_mask_time, size=(num_masks_time, 2))
        for m, (t, w) in enumerate(ts):
            if xlen - t <= 0:
                continue
            t_zero = py_rng.randrange(0, xlen - t)
            if t == 0:
                continue
            spans[b, num_masks_freq + m] = (t_zero, min(t_zero + w, int(xlen)))
    return spans
