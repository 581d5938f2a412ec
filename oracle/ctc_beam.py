"""CPU oracle, part 4: CTC prefix beam search with LM shallow fusion for CTC-only models.
TEST INFRASTRUCTURE ONLY (see oracle/model.py): nothing on the product path may import this.

Restates asr/modeling/decoders/ctc.py:203-344 (`_beam_search`) and :372-397 (`_merge_ctc_paths`) of
/root/reference, including the reference's quirks:
  * `score_lm` is a running variable inside one beam's candidate loop, so the LM score of candidate k
    also contains the LM scores of the candidates before it (ctc.py:309-310);
  * the length bonus is len_weight * (len(strip_eos(hyp)) + 1) (ctc.py:308);
  * merged paths keep the FIRST path's score_lm / score_len (ctc.py:388-393);
  * non-extended copies keep the old LM score; the LM is run on every beam at every frame when
    lm_weight > 0 (ctc.py:241-260), on 0-padded prefixes.
Pinned by tests/test_oracle_ctc_beam.py against tests/golden/ctcbeam_tiny.npz (reference outputs).
"""
import numpy as np

LOG_0 = -1e10  # decoders/ctc.py:23


def _merge(beams):
    """ctc.py:372-397: paths with the same label sequence are folded with logaddexp"""
    merged = {}
    for beam in beams:
        key = " ".join(map(str, beam["hyp"]))
        if key in merged:
            m = merged[key]
            m["p_b"] = np.logaddexp(m["p_b"], beam["p_b"])
            m["p_nb"] = np.logaddexp(m["p_nb"], beam["p_nb"])
            m["score_asr"] = np.logaddexp(m["score_asr"], beam["score_asr"])
            m["score"] = m["score_asr"] + m["score_lm"] + m["score_len"]  # score_lm / score_len: first path's
        else:
            merged[key] = beam
    return list(merged.values())


def ctc_prefix_beam_search(log_probs, blank_id, eos_id, beam_width, len_weight=0.0, lm_predict=None, lm_weight=0.0):
    """log_probs: float array [T, V] (log_softmax of the CTC head for ONE utterance).
    lm_predict(hyps_batch int64 [nb, N] (0-padded), lens [nb]) -> array [nb, V] of next-token log-probs.
    -> (hyps, scores), best first; every hyp starts with <eos> like the reference's."""
    T, V = log_probs.shape
    beams = [dict(hyp=[eos_id], score=0.0, p_b=0.0, p_nb=LOG_0, score_asr=0.0, score_lm=0.0, score_len=0.0)]
    k = min(beam_width, V)
    for t in range(T):
        row = log_probs[t]
        # torch.topk(sorted=True): descending values (ties do not occur in the pinned vectors)
        v_topk = np.argsort(-row, kind="stable")[:k]
        if lm_weight > 0:
            n = max(len(b["hyp"]) for b in beams)
            batch = np.zeros((len(beams), n), dtype=np.int64)
            for i, b in enumerate(beams):
                batch[i, : len(b["hyp"])] = b["hyp"]
            lm_lp = np.asarray(lm_predict(batch, np.array([len(b["hyp"]) for b in beams])))
        new_beams = []
        for bi, beam in enumerate(beams):
            hyp, p_b, p_nb = beam["hyp"], beam["p_b"], beam["p_nb"]
            score_lm, score_len = beam["score_lm"], beam["score_len"]
            # case 1: not extended
            new_p_b = np.logaddexp(p_b + float(row[blank_id]), p_nb + float(row[blank_id]))
            new_p_nb = p_nb + float(row[hyp[-1]]) if len(hyp) > 1 else LOG_0
            score_asr = np.logaddexp(new_p_b, new_p_nb)
            new_beams.append(dict(hyp=hyp, score=score_asr + score_lm + score_len, p_b=new_p_b, p_nb=new_p_nb,
                                  score_asr=score_asr, score_lm=score_lm, score_len=score_len))
            # case 2: extended by each of the frame's top-k labels
            new_p_b = LOG_0
            for v in v_topk:
                v = int(v)
                p_t = float(row[v])
                if v == blank_id:
                    continue
                v_prev = hyp[-1] if len(hyp) > 1 else None
                new_p_nb = p_b + p_t if v == v_prev else np.logaddexp(p_b + p_t, p_nb + p_t)
                score_asr = np.logaddexp(new_p_b, new_p_nb)
                score_len = len_weight * (len([x for x in hyp if x != eos_id]) + 1)
                if lm_weight > 0:
                    score_lm += lm_weight * float(lm_lp[bi, v])  # running sum over the candidates (reference quirk)
                new_beams.append(dict(hyp=hyp + [v], score=score_asr + score_lm + score_len, p_b=new_p_b,
                                      p_nb=new_p_nb, score_asr=score_asr, score_lm=score_lm, score_len=score_len))
        beams = sorted(_merge(new_beams), key=lambda x: x["score"], reverse=True)[:beam_width]
    return [b["hyp"] for b in beams], [float(b["score"]) for b in beams]
