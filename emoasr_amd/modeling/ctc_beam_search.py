"""CTC prefix beam search with LM shallow fusion for CTC-only models -- the decode path of
asr/modeling/decoders/ctc.py:203-344 (`CTCDecoder._beam_search`) and :372-397 (`_merge_ctc_paths`).

Split of work: the acoustic side runs once per utterance on the GPU (output projection, log-softmax
and the per-frame top-k, all frames in one launch each: emoasr_gemm_nt / emoasr_log_softmax /
emoasr_topk) and comes to the host in one copy; the Transformer LM scores the NEW prefixes of a frame
in one batched device call (LM.predict, as the reference batches them with pad_sequence, ctc.py:241-260),
its rows stay on the device while their prefix is live and a frame brings only its k candidate columns to
the host; the prefix bookkeeping itself -- ~110 candidates per frame: extend, merge, stable sort, prune --
is native host code in IEEE doubles, like the reference's python floats (csrc/ctc_beam_host.hip:
emoasr_ctc_beam_step, 20 us per frame).  EMOASR_CTC_BEAM_NATIVE=0 runs the same bookkeeping as the Python
loop below (bit-identical results; tests/test_ctc_beam_gpu.py compares the two).

Reference behaviour kept on purpose (it decides which prefixes survive):
  * the LM score of the k-th candidate extension of a prefix also contains the LM scores of the
    candidates tried before it (`score_lm +=` inside the candidate loop, ctc.py:309-310);
  * length bonus = len_weight * (number of non-<eos> tokens of the PARENT prefix + 1) (ctc.py:308);
  * when two paths reach the same prefix, probabilities are merged but the LM / length scores of the
    first path are kept (ctc.py:388-393);
  * hypotheses start with <eos> (the LM's BOS) and scores are returned best first.
"""
import math
import os

import numpy as np
import torch

from .. import ops
from .functions import _engine_of

NEG = -1e10  # LOG_0 of decoders/ctc.py:23


def _lse2(a, b):
    m = a if a > b else b
    return m + math.log(math.exp(a - m) + math.exp(b - m))


class _Prefix:
    __slots__ = ("toks", "p_b", "p_nb", "asr", "lm", "len_bonus", "n_plain")

    def __init__(self, toks, p_b, p_nb, asr, lm, len_bonus, n_plain):
        self.toks, self.p_b, self.p_nb, self.asr, self.lm, self.len_bonus = toks, p_b, p_nb, asr, lm, len_bonus
        self.n_plain = n_plain  # tokens that are not <eos>

    @property
    def total(self):
        return self.asr + self.lm + self.len_bonus


def _search_native(logp_dev, top_dev, T, V, k, blank, eos, beam_width, len_weight, lm, lm_weight):
    """the frame loop over csrc/ctc_beam_host.hip: per frame one C call (bookkeeping), and -- with an LM -- one batched LM call for
    the prefixes that are new plus one [live, k] gather of their cached rows (the frame's only device-to-host copy)"""
    import ctypes
    from .. import lib
    L = lib.load()
    L.emoasr_ctc_beam_new.restype = ctypes.c_void_p
    L.emoasr_ctc_beam_new.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double]
    L.emoasr_ctc_beam_free.argtypes = [ctypes.c_void_p]
    L.emoasr_ctc_beam_free.restype = None
    L.emoasr_ctc_beam_step.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] + [ctypes.c_void_p] * 3
    L.emoasr_ctc_beam_scores.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    use_lm = lm is not None and lm_weight > 0
    logp = logp_dev.cpu().numpy()                      # f32 [T, V]: one D2H for the whole utterance (rows read as doubles in C)
    top = np.ascontiguousarray(top_dev.cpu().numpy().astype(np.int32))
    h = L.emoasr_ctc_beam_new(int(beam_width), int(blank), int(eos), float(len_weight), float(lm_weight if use_lm else 0.0))
    assert h, "emoasr_ctc_beam_new failed"
    try:
        parent = np.zeros(beam_width, dtype=np.int32)
        tok = np.zeros(beam_width, dtype=np.int32)
        live = [(eos,)]
        dev = logp_dev.device
        if use_lm:
            nslot = 2 * beam_width + 2
            cache = torch.empty(nslot, V, device=dev, dtype=torch.float32)   # LM rows of the live prefixes
            slot_of, free = {}, list(range(nslot))
            top_long = top_dev.long()
        for t in range(T):
            lm_ptr = None
            if use_lm:
                need = [p for p in live if p not in slot_of]
                if need:
                    n = max(len(p) for p in need)
                    batch = torch.zeros(len(need), n, dtype=torch.int64)   # 0-padded like pad_sequence (ctc.py:243-246)
                    for i, p in enumerate(need):
                        batch[i, : len(p)] = torch.tensor(p)
                    rows, _ = lm.predict(batch, [len(p) for p in need])
                    ids = []
                    for p in need:
                        slot_of[p] = free.pop()
                        ids.append(slot_of[p])
                    cache.index_copy_(0, torch.tensor(ids, device=dev), rows.float())
                sl = torch.tensor([slot_of[p] for p in live], device=dev)
                vals = cache.index_select(0, sl).index_select(1, top_long[t]).cpu().numpy().astype(np.float64)   # [live, k]
                vals = np.ascontiguousarray(vals)
                lm_ptr = vals.ctypes.data
            n = L.emoasr_ctc_beam_step(h, logp[t].ctypes.data, top[t].ctypes.data, int(k), lm_ptr, parent.ctypes.data, tok.ctypes.data)
            if n < 0:
                raise lib.EmoasrHipError("emoasr_ctc_beam_step failed: " + L.emoasr_last_error().decode())
            live = [live[parent[i]] + ((int(tok[i]),) if tok[i] >= 0 else ()) for i in range(n)]
            if use_lm:
                keep = set(live)
                for key in [q for q in slot_of if q not in keep]:
                    free.append(slot_of.pop(key))
        scores = np.zeros(len(live), dtype=np.float64)
        L.emoasr_ctc_beam_scores(h, scores.ctypes.data)
        return [list(p) for p in live], [float(v) for v in scores]
    finally:
        L.emoasr_ctc_beam_free(h)


def ctc_prefix_beam_search(dec, eouts, elens, beam_width, len_weight=0.0, lm=None, lm_weight=0.0):
    """-> (hyps, scores, logits) for ONE utterance (the reference asserts batch size 1, ctc.py:212)."""
    assert eouts.shape[0] == 1, "CTC beam search decodes one utterance at a time (ctc.py:212)"
    eng = _engine_of(dec)
    blank, eos, V = dec.blank_id, dec.eos_id, dec.vocab_size
    with torch.no_grad():
        logits = eng.head_logits(eouts, getattr(dec, "_prefix", "decoder") + ".output")                      # [1, T, V]
        T = logits.shape[1]
        logp_dev = ops.log_softmax(logits.view(T, V))        # f32 [T, V]
        k = min(beam_width, V)
        _, top_dev, _ = ops.topk(logp_dev, k)                # ids sorted by descending score, per frame
        if os.environ.get("EMOASR_CTC_BEAM_NATIVE", "1") != "0":
            hyps, scores = _search_native(logp_dev, top_dev, T, V, k, blank, eos, beam_width, len_weight, lm, lm_weight)
            return hyps, scores, logits
        logp = logp_dev.cpu().numpy().astype(np.float64)     # one D2H for the whole utterance
        top = top_dev.cpu().numpy()
    use_lm = lm is not None and lm_weight > 0
    lm_cache, lm_rows = os.environ.get("EMOASR_CTC_LM_CACHE", "1") != "0", {}
    live = [_Prefix((eos,), 0.0, NEG, 0.0, 0.0, 0.0, 0)]
    for t in range(T):
        row = logp[t]
        lp_blank = float(row[blank])
        cands = [(int(v), float(row[v])) for v in top[t] if int(v) != blank]
        if use_lm:
            # The reference scores EVERY live prefix with the LM at every frame (ctc.py:241-260) -- the same prefix again and again
            # while it survives.  The LM's next-token row is a function of the prefix alone, so each distinct prefix is scored
            # ONCE (the frame it first survives) and its row kept while it is live: a frame costs an LM call only for prefixes
            # that are new, in one batch (rows of a batched call do not depend on their neighbours).  EMOASR_CTC_LM_CACHE=0:
            # the reference's recomputation.
            need = [p for p in live if p.toks not in lm_rows] if lm_cache else list(live)
            if need:
                n = max(len(p.toks) for p in need)
                batch = torch.zeros(len(need), n, dtype=torch.int64)  # 0-padded like pad_sequence (ctc.py:243-246)
                for i, p in enumerate(need):
                    batch[i, : len(p.toks)] = torch.tensor(p.toks)
                rows, _ = lm.predict(batch, [len(p.toks) for p in need])
                rows = rows.cpu().numpy().astype(np.float64)
                for i, p in enumerate(need):
                    lm_rows[p.toks] = rows[i]
            lm_lp = [lm_rows[p.toks] for p in live]
            if lm_cache:
                keep = {p.toks for p in live}
                for key in [k for k in lm_rows if k not in keep]:
                    del lm_rows[key]
            else:
                lm_rows.clear()
        table, order = {}, []  # prefix -> _Prefix, in first-seen order (dict merge of ctc.py:374-395)

        def put(q):
            old = table.get(q.toks)
            if old is None:
                table[q.toks] = q
                order.append(q)
            else:  # same label sequence reached twice: fold the path probabilities only
                old.p_b, old.p_nb, old.asr = _lse2(old.p_b, q.p_b), _lse2(old.p_nb, q.p_nb), _lse2(old.asr, q.asr)

        for i, p in enumerate(live):
            last = p.toks[-1] if len(p.toks) > 1 else None
            # stay on the same prefix: blank, or a repeat of its last label
            stay_b = _lse2(p.p_b + lp_blank, p.p_nb + lp_blank)
            stay_nb = p.p_nb + float(row[last]) if last is not None else NEG
            put(_Prefix(p.toks, stay_b, stay_nb, _lse2(stay_b, stay_nb), p.lm, p.len_bonus, p.n_plain))
            # extend by each of the frame's top-k labels
            lm_run = p.lm
            bonus = len_weight * (p.n_plain + 1)
            for v, lp_v in cands:
                ext_nb = p.p_b + lp_v if v == last else _lse2(p.p_b + lp_v, p.p_nb + lp_v)
                if use_lm:
                    lm_run += lm_weight * float(lm_lp[i][v])
                put(_Prefix(p.toks + (v,), NEG, ext_nb, _lse2(NEG, ext_nb), lm_run, bonus,
                            p.n_plain + (0 if v == eos else 1)))
        order.sort(key=lambda q: q.total, reverse=True)  # stable, like sorted() in ctc.py:338
        live = order[:beam_width]
    return [list(p.toks) for p in live], [p.total for p in live], logits
