"""Joint CTC / attention beam search with Transformer-LM shallow fusion on the GPU.

Behaviour follows TransformerDecoder.decode (asr/modeling/decoders/transformer.py:161-294)
including its quirks (see oracle/decoder.py:joint_beam_search): the LM term already sits inside
`scores_att` when the CTC re-scoring adds it again; finished hypotheses get
`score + len_weight * len(hyp incl. sos/eos)`; empty hypotheses are dropped; the search stops as
soon as `beam_width` results exist.

Device side per output step (all beams batched): decoder logits of the last position, LM
log-probabilities, log-softmax + fusion, top-k candidate selection, CTC prefix scores for every
(beam, candidate) with the scorer states kept on the device.  Host side: one small D2H per step
(candidate ids/scores), then the reference's own list bookkeeping (sort, prune, finish).
"""
import os

import numpy as np
import torch

from .. import ops
from ..engine import h2d_i32
from .functions import _engine_of

CTC_BEAM_WIDTH_RATIO = 1.5


def joint_beam_search(dec, eouts, elens, beam_width, len_weight=0, lm=None, lm_weight=0, decode_ctc_weight=0):
    eng = _engine_of(dec)
    assert eouts.shape[0] == 1, "beam search decodes one utterance at a time (decoders/transformer.py:181)"
    V, eos, blank = dec.vocab_size, dec.eos_id, dec.blank_id
    # the whole step on the device (K / V caches, device-side bookkeeping: one C-ABI call and one flag per step);
    # EMOASR_DEVICE_BEAM=0 keeps the host bookkeeping below (same results)
    if (os.environ.get("EMOASR_DEVICE_BEAM", "1") != "0" and os.environ.get("EMOASR_CPP_DECODE", "1") != "0"
            and beam_width <= 32 and min(V, int(beam_width * CTC_BEAM_WIDTH_RATIO)) <= 32
            and (lm is None or lm_weight <= 0 or hasattr(lm, "predict_device"))):
        from .beam_search_device import joint_beam_search_device
        return joint_beam_search_device(dec, eouts, elens, beam_width, len_weight, lm, lm_weight, decode_ctc_weight)
    dev = eouts.device
    T = eouts.shape[1]
    use_ctc = decode_ctc_weight > 0
    use_lm = lm is not None and lm_weight > 0
    lam, mu = float(decode_ctc_weight), float(lm_weight)
    with torch.no_grad():
        if use_ctc:
            ctc_logits = eng.head_logits(eouts, "decoder.ctc.output")
            x = ops.log_softmax(ctc_logits.view(T, V))
            init_state = ops.ctc_prefix_init(x, blank)
            cw = min(V, int(beam_width * CTC_BEAM_WIDTH_RATIO))
        # C++ step runtime (csrc/decode_rt.hip): one C-ABI call per network per output step, cross-attention
        # K / V of the memory projected once per utterance.  EMOASR_CPP_DECODE=0: per-kernel Python sequencing.
        rt = lmrt = None
        if os.environ.get("EMOASR_CPP_DECODE", "1") != "0":
            from ..decode_rt import DecoderStepRuntime, LMStepRuntime
            rt = getattr(eng, "_dec_rt", None)
            if rt is None:
                rt = eng._dec_rt = DecoderStepRuntime(eng)
            rt.begin(eouts, beam_width)
            if use_lm and hasattr(lm, "predict_device"):
                lmrt = getattr(lm, "_step_rt", None)
                if lmrt is None:
                    lmrt = lm._step_rt = LMStepRuntime(lm)
        side = None
        if lmrt is not None and os.environ.get("EMOASR_DECODE_STREAMS", "1") != "0":
            side = getattr(eng, "_lm_stream", None)
            if side is None:
                side = eng._lm_stream = torch.cuda.Stream(device=dev)
        beams = [dict(hyp=[eos], score=0.0, score_ctc=np.float32(0.0), parent=0, pcand=0)]
        prev_states = None
        results = []
        for i in range(dec.max_decode_ylen):
            nb = len(beams)
            ys_in = torch.tensor([b["hyp"] for b in beams], dtype=torch.int64)
            if rt is not None:
                if lmrt is not None and side is not None:
                    # the LM and the decoder are independent chains of ~90 small kernels each: run the LM on a
                    # second stream so the two chains overlap (both are bound by the per-kernel dispatch floor)
                    main = torch.cuda.current_stream()
                    side.wait_stream(main)
                    with torch.cuda.stream(side), ops.stream_scope():
                        lm_lp = lmrt.step(ys_in)
                    with ops.stream_scope():
                        last = rt.step(ys_in)                                        # [nb, V]
                    main.wait_stream(side)
                    lm_lp.record_stream(main)
                else:
                    with ops.stream_scope():
                        last = rt.step(ys_in)
                        lm_lp = (lmrt.step(ys_in) if lmrt is not None else lm.predict_device(ys_in, [i + 1] * nb)) if use_lm else None
            else:
                mem = eouts.expand(nb, T, eouts.shape[2]).contiguous()
                el = h2d_i32([T] * nb, dev)
                logits, _ = eng.dec_forward(mem, el, ys_in, [i] * nb, False, False)  # [nb, i+1, V]
                last = logits[:, i]
                lm_lp = lm.predict_device(ys_in, [i + 1] * nb) if use_lm else None
            scores_pre = ops.log_softmax(last, add=lm_lp, mu=mu)  # = scores_att (+ lm: the in-place alias quirk)
            if use_ctc:
                vals, cands, lm_at = ops.topk(scores_pre, cw, aux=lm_lp)
                last_tok = h2d_i32([b["hyp"][-1] for b in beams], dev)
                out_len = h2d_i32([len(b["hyp"]) - 1 for b in beams], dev)
                parent = h2d_i32([b["parent"] for b in beams], dev)
                pcand = h2d_i32([b["pcand"] for b in beams], dev)
                log_psi, states = ops.ctc_prefix_score(x, cands, last_tok, out_len, blank, eos, prev_states, parent, pcand,
                                                       init_state)
                vals_h, cands_h, psi_h = vals.cpu().numpy(), cands.cpu().numpy(), log_psi.cpu().numpy()
                lm_h = lm_at.cpu().numpy() if use_lm else None
                prev_states = states
            else:
                vals, idx, _ = ops.topk(scores_pre, beam_width)
                vals_h, idx_h = vals.cpu().numpy(), idx.cpu().numpy()
            new_beams = []
            for m, beam in enumerate(beams):
                if use_ctc:
                    sc = np.float32(1 - lam) * vals_h[m] + np.float32(lam) * (psi_h[m] - beam["score_ctc"])
                    if use_lm:
                        sc = sc + np.float32(mu) * lm_h[m]
                    order = np.argsort(-sc, kind="stable")[:beam_width]
                    for j in order:
                        new_beams.append(dict(score=beam["score"] + float(sc[j]), hyp=beam["hyp"] + [int(cands_h[m, j])],
                                              score_ctc=psi_h[m, j], parent=m, pcand=int(j)))
                else:
                    for j in range(beam_width):
                        new_beams.append(dict(score=beam["score"] + float(vals_h[m, j]), hyp=beam["hyp"] + [int(idx_h[m, j])],
                                              score_ctc=np.float32(0.0), parent=m, pcand=0))
            beams = sorted(new_beams, key=lambda b: b["score"], reverse=True)[:beam_width]
            alive = []
            for beam in beams:
                if beam["hyp"][-1] == eos:
                    hyp = [t for t in beam["hyp"] if t != eos]
                    if len(hyp) < 1:
                        continue
                    results.append(dict(hyp=hyp, score=beam["score"] + len_weight * len(beam["hyp"])))
                    if len(results) >= beam_width:
                        break
                else:
                    alive.append(beam)
            if len(results) >= beam_width or not alive:
                break
            beams = alive
    results = sorted(results, key=lambda r: r["score"], reverse=True)
    return [r["hyp"] for r in results], [r["score"] for r in results], None, None
