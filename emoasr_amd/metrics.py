"""Word / character error rate (asr/metrics.py:20-175): Levenshtein alignment with the reference's
backtrace preference (correct, then insertion, then substitution, then deletion), so the reported
D / S / I split is the same as the reference's, not just the distance."""
import numpy as np


def compute_wer(hyp, ref, cer=False):
    """-> (wer in %, dict(wer, n_sub, n_ins, n_del, n_ref, error_list)); an empty hypothesis is scored as
    one dummy word that matches nothing (metrics.py:21-23)."""
    hyp = list(hyp) if len(hyp) else ["<dummy>"]
    ref = list(ref)
    if cer:
        hyp, ref = list("".join(hyp)), list("".join(ref))
    R, H = len(ref), len(hyp)
    d = np.zeros((R + 1, H + 1), dtype=np.int64)
    d[0, :] = np.arange(H + 1)
    d[:, 0] = np.arange(R + 1)
    for i in range(1, R + 1):
        ri = ref[i - 1]
        for j in range(1, H + 1):
            d[i, j] = d[i - 1, j - 1] if ri == hyp[j - 1] else 1 + min(d[i - 1, j - 1], d[i, j - 1], d[i - 1, j])
    ops, x, y = [], R, H
    while x or y:
        if x and y and d[x, y] == d[x - 1, y - 1] and ref[x - 1] == hyp[y - 1]:
            ops.append("C"); x -= 1; y -= 1
        elif y and d[x, y] == d[x, y - 1] + 1:
            ops.append("I"); y -= 1
        elif x and y and d[x, y] == d[x - 1, y - 1] + 1:
            ops.append("S"); x -= 1; y -= 1
        else:
            ops.append("D"); x -= 1
    ops.reverse()
    n_sub, n_ins, n_del = ops.count("S"), ops.count("I"), ops.count("D")
    assert int(d[R, H]) == n_sub + n_ins + n_del
    wer = 100.0 * int(d[R, H]) / R
    return wer, {"wer": wer, "n_sub": n_sub, "n_ins": n_ins, "n_del": n_del, "n_ref": R, "error_list": ops}


def _total(pairs, cer):
    tot = {"n_sub": 0, "n_ins": 0, "n_del": 0, "n_ref": 0}
    for hyp, ref in pairs:
        _, w = compute_wer(hyp, ref, cer=cer)
        for k in tot:
            tot[k] += w[k]
    wer = 100.0 * (tot["n_sub"] + tot["n_ins"] + tot["n_del"]) / tot["n_ref"]
    return wer, dict(tot, wer=wer)


def compute_wers(hyps, refs, vocab=None, cer=False):
    """corpus-level WER over id (with vocab: subword -> word) or word sequences (metrics.py:108-130)"""
    if vocab is not None:
        hyps, refs = [vocab.ids2words(h) for h in hyps], [vocab.ids2words(r) for r in refs]
    return _total(zip(hyps, refs), cer)


def compute_wers_df(dfhyp, dfref=None, cer=False):
    """result TSV (columns text, reftext) or hypothesis + reference tables joined on utt_id
    (metrics.py:133-175); missing / NaN hypotheses count as empty"""
    def words(v):
        return v.split() if isinstance(v, str) else []
    if dfref is None:
        return _total(((words(r.text), words(r.reftext)) for r in dfhyp.itertuples()), cer)
    id2hyp = {r.utt_id: words(r.text) for r in dfhyp.itertuples()}
    return _total(((id2hyp.get(r.utt_id, []), words(r.text)) for r in dfref.itertuples()), cer)


def wer_summary(wer, w, cer=False):
    return f"{'CER' if cer else 'WER'}: {wer:.2f} [D={w['n_del']:d}, S={w['n_sub']:d}, I={w['n_ins']:d}, N={w['n_ref']:d}]"
