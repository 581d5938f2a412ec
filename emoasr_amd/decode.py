"""Decoding driver with the reference's protocol (asr/test_asr.py:36-121,226-263): one utterance per step,
result rows `[utt_id, token_id, text, reftext]` (n-best: `[utt_id, score, token_id, text, reftext]`), and
the real-time-factor measurement (wall time of `test` over `num_samples` utterances / audio seconds, mean
over repeats).

    rows = test(model, dataloader, vocab, beam_width, len_weight, decode_ctc_weight, decode_phone, lm, lm_weight, device)
    runtime_utt, rtf = measure_rtf(model, dataloader, vocab, ..., num_samples=20, num_repeats=5)

Audio duration comes from the utterance id (`..._<start>_<end>` in units of 1/wavtime_factor s, as in the
reference's corpora) or, when the id carries no times, from the frame count (10 ms per input frame).
"""
import logging
import re
import time

import numpy as np
import torch


def strip_eos(tokens, eos_id):
    """utils/converters.py:29-30"""
    return [t for t in tokens if t != eos_id]


def ints2str(ints):
    """utils/converters.py:13-14"""
    return " ".join(str(i) for i in ints)


def test_step(model, data, beam_width, len_weight, decode_ctc_weight, decode_phone, lm, lm_weight, device):
    """test_asr.py:36-60: decodes the (single-utterance) batch -> (utt_id, hyps, scores, reftext)"""
    utt_id = data["utt_ids"][0]
    reftext = data["ptexts"][0] if decode_phone else data["texts"][0]
    hyps, scores, _, _ = model.decode(data["xs"].to(device), data["xlens"], beam_width, len_weight, lm=lm,
                                      lm_weight=lm_weight, decode_ctc_weight=decode_ctc_weight, decode_phone=decode_phone)
    return utt_id, hyps, scores, reftext


def test(model, dataloader, vocab, beam_width, len_weight, decode_ctc_weight, decode_phone, lm, lm_weight, device,
         eos_id=2, num_samples=-1, sample_utt_id=None, nbest=False):
    """test_asr.py:63-121 -> rows.  An utterance without any hypothesis gives token_id None and an empty text."""
    from .hostenv import respect_cpu_quota
    respect_cpu_quota()   # (an oversized CPU pool under a cgroup quota freezes the decoding thread for 20-50 ms at a time)
    rows = []
    n = len(dataloader) if hasattr(dataloader, "__len__") else -1
    # evaluation: the weights are constant for the whole loop (test_asr.py:63 runs under model.eval() / torch.no_grad())
    arenas = []
    for m in (model, lm):
        eng = m.engine() if m is not None and hasattr(m, "engine") else None
        if eng is not None and hasattr(eng, "arena"):
            eng.ensure_bound()
            arenas.append(eng.arena)
    for a in arenas:
        a.hold_shadow(True)
    try:
        return _test_rows(model, dataloader, vocab, beam_width, len_weight, decode_ctc_weight, decode_phone, lm, lm_weight, device,
                          eos_id, num_samples, sample_utt_id, nbest, rows, n)
    finally:
        for a in arenas:
            a.hold_shadow(False)


def _test_rows(model, dataloader, vocab, beam_width, len_weight, decode_ctc_weight, decode_phone, lm, lm_weight, device, eos_id,
               num_samples, sample_utt_id, nbest, rows, n):
    for i, data in enumerate(dataloader):
        if num_samples > 0 and (i + 1) > num_samples:
            return rows
        if sample_utt_id is not None and sample_utt_id != data["utt_ids"][0]:
            continue
        utt_id, hyps, scores, reftext = test_step(model, data, beam_width, len_weight, decode_ctc_weight, decode_phone,
                                                  lm, lm_weight, device)
        text = ""
        if nbest:
            for hyp, score in zip(hyps, scores):
                ids = strip_eos(hyp, eos_id)
                rows.append([utt_id, score, ints2str(ids), vocab.ids2text(ids), reftext])
            text = vocab.ids2text(strip_eos(hyps[0], eos_id))
        else:
            if len(hyps) < 1:
                token_id, text = None, ""
                logging.warning(f"cannot decode {utt_id}")
            else:
                ids = strip_eos(hyps[0], eos_id)
                token_id, text = ints2str(ids), vocab.ids2text(ids)
            rows.append([utt_id, token_id, text, reftext])
        logging.debug(f"{utt_id}({(i + 1):d}/{n:d}): {text}")
    return rows


def wavtime_of(utt_id, wavtime_factor=1000.0):
    """seconds of audio from `..._<start>_<end>` / `...-<start>-<end>` (test_asr.py:251-254); None if absent"""
    parts = re.split("_|-", utt_id)
    try:
        return (int(parts[-1]) - int(parts[-2])) / wavtime_factor
    except (ValueError, IndexError):
        return None


def measure_rtf(model, dataloader, vocab, beam_width, len_weight, decode_ctc_weight, decode_phone, lm, lm_weight, device,
                eos_id=2, num_samples=20, num_repeats=5, wavtime_factor=1000.0, frame_seconds=0.010):
    """test_asr.py:226-263 -> (mean runtime per utterance [s], mean RTF).  The device is synchronised before
    each clock read, so asynchronous launches are inside the measured time."""
    frames = {}

    def recording(loader):  # the loader is walked INSIDE the timed region, feature loading included (test_asr.py:231-240)
        for d in loader:
            frames[d["utt_ids"][0]] = int(d["xlens"][0])
            yield d

    runtimes, rtfs = [], []
    for j in range(num_repeats):
        torch.cuda.synchronize()
        t0 = time.time()
        rows = test(model, recording(dataloader), vocab, beam_width, len_weight, decode_ctc_weight, decode_phone, lm, lm_weight,
                    device, eos_id=eos_id, num_samples=num_samples)
        torch.cuda.synchronize()
        runtime = time.time() - t0
        wavtime = 0.0
        for r in rows:
            w = wavtime_of(r[0], wavtime_factor)
            wavtime += w if w is not None and w > 0 else frames[r[0]] * frame_seconds
        runtimes.append(runtime / max(num_samples, 1))
        rtfs.append(runtime / wavtime)
        logging.info(f"Run {(j + 1):d} | runtime: {runtimes[-1]:.5f}sec / utt, wavtime: {wavtime:.5f}sec | RTF: {rtfs[-1]:.5f}")
    logging.info(f"Averaged runtime {np.mean(runtimes):.5f}sec, RTF {np.mean(rtfs):.5f} on {torch.device(device).type}")
    return float(np.mean(runtimes)), float(np.mean(rtfs))


def save_results(rows, path, nbest=False, phone=False):
    """test_asr.py:296-313: result TSV; for 1-best results the WER summary is computed and written as the
    file's leading `# ...` comment -> (wer, wer_info) or None"""
    from .datasets import write_results_tsv
    from .metrics import compute_wers_df
    import pandas as pd
    cols = ["utt_id", "score_asr", "token_id", "text", "reftext"] if nbest else ["utt_id", "token_id", "text", "reftext"]
    df = pd.DataFrame(rows, columns=cols)
    if nbest:
        df.to_csv(path, sep="\t", index=False)
        return None
    wer, w = compute_wers_df(df)
    info = f"{'PER' if phone else 'WER'}: {wer:.2f} [D={w['n_del']:d}, S={w['n_sub']:d}, I={w['n_ins']:d}, N={w['n_ref']:d}]"
    write_results_tsv(path, [dict(zip(cols, ["" if v is None else v for v in r])) for r in rows], comment=info)
    return wer, info
