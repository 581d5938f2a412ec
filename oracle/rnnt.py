"""CPU oracle, part 3: RNN-Transducer decoder (LSTM prediction network, joint network, transducer
loss, time-synchronous greedy decoding).  TEST INFRASTRUCTURE ONLY (see oracle/model.py).

Citations (file:line in /root/reference):
  RNNTDecoder.forward / joint / recurrency / _greedy   asr/modeling/decoders/rnn_transducer.py:81-240

PARITY UNPINNED for the loss value: the reference calls the third-party CUDA package `warp_rnnt`
(PyPI warp-rnnt, github 1ytic/warp-rnnt; version unpinned, not installed, rnn_transducer.py:106-115).
`rnnt_loss` below restates the published transducer forward algorithm (Graves 2012, eq. 16-18) with
that call's semantics (inputs already log-softmaxed, no frame averaging, blank index, reduction
"mean" over the batch) and is validated by brute-force path enumeration in
tests/test_oracle_rnnt.py; everything around the loss (LSTM, joint, auxiliary CTC, greedy decode) IS
pinned to the reference by tests/golden/l4_tiny.npz, generated with this function plugged in as
`warp_rnnt.rnnt_loss`.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .model import ctc_loss, linear


def rnnt_nll(log_probs, labels, flens, llens, blank=0):
    """per-utterance -log p(y|x).  log_probs [B,T,U+1,V] (normalised), labels [B,U] int."""
    B = log_probs.shape[0]
    out = []
    for b in range(B):
        T, U = int(flens[b]), int(llens[b])
        lp = log_probs[b]
        ninf = lp.new_full((), float("-inf"))
        alpha = [[None] * (U + 1) for _ in range(T)]
        for t in range(T):
            for u in range(U + 1):
                if t == 0 and u == 0:
                    alpha[t][u] = lp.new_zeros(())
                    continue
                stay = alpha[t - 1][u] + lp[t - 1, u, blank] if t > 0 else ninf
                emit = alpha[t][u - 1] + lp[t, u - 1, labels[b, u - 1]] if u > 0 else ninf
                alpha[t][u] = torch.logaddexp(stay, emit)
        out.append(-(alpha[T - 1][U] + lp[T - 1, U, blank]))
    return torch.stack(out)


def rnnt_loss(log_probs, labels, frames_lengths, labels_lengths, average_frames=False, reduction=None, blank=0,
              gather=False):
    """drop-in for warp_rnnt.rnnt_loss (the call at rnn_transducer.py:106-115)"""
    assert not average_frames and not gather
    nll = rnnt_nll(log_probs, labels.long(), frames_lengths, labels_lengths, blank)
    if reduction == "mean":
        return nll.mean()
    if reduction == "sum":
        return nll.sum()
    return nll


def lstm_layer(sd, name, x, h0, c0):
    """single-layer batch-first LSTM, gate order i, f, g, o (torch.nn.LSTM) -> (ys [B,L,H], (h, c))"""
    w_ih, w_hh = sd[name + ".weight_ih_l0"], sd[name + ".weight_hh_l0"]
    b = sd[name + ".bias_ih_l0"] + sd[name + ".bias_hh_l0"]
    H = w_hh.shape[1]
    h, c = h0, c0
    ys = []
    for t in range(x.shape[1]):
        gates = F.linear(x[:, t], w_ih) + F.linear(h, w_hh) + b
        i, f, g, o = gates[:, :H], gates[:, H:2 * H], gates[:, 2 * H:3 * H], gates[:, 3 * H:]
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        ys.append(h)
    return torch.stack(ys, 1), (h, c)


def recurrency(sd, cfg, ys_in, state=None, prefix="decoder"):
    """prediction network: embedding + dec_num_layers LSTMs (dropout not modelled)"""
    x = F.embedding(ys_in, sd[prefix + ".embed.weight"])
    B, n = x.shape[0], cfg.dec_num_layers
    H = cfg.dec_hidden_size
    if state is None:
        state = (x.new_zeros(n, B, H), x.new_zeros(n, B, H))
    hs, cs = [], []
    for l in range(n):
        x, (h, c) = lstm_layer(sd, f"{prefix}.rnns.{l}", x, state[0][l], state[1][l])
        hs.append(h)
        cs.append(c)
    return x, (torch.stack(hs), torch.stack(cs))


def joint(sd, eouts, douts, prefix="decoder"):
    """logits [B,T,U,V] = W_out tanh(W_enc e_t + W_dec g_u)"""
    h = torch.tanh(linear(sd, prefix + ".w_enc", eouts).unsqueeze(2) + linear(sd, prefix + ".w_dec", douts).unsqueeze(1))
    return linear(sd, prefix + ".output", h)


def rnnt_decoder_forward(sd, cfg, eouts, elens, ys, ylens, ys_in, prefix="decoder"):
    douts, _ = recurrency(sd, cfg, ys_in, None, prefix)
    logits = joint(sd, eouts, douts, prefix)
    loss_rnnt = rnnt_loss(torch.log_softmax(logits, -1), ys, elens, ylens, reduction="mean", blank=cfg.blank_id)
    loss, ld = loss_rnnt, {"loss_rnnt": loss_rnnt}
    if cfg.mtl_ctc_weight > 0:
        lc = ctc_loss(linear(sd, prefix + ".ctc.output", eouts), ys, elens, ylens, cfg.blank_id)
        loss = loss + cfg.mtl_ctc_weight * lc
        ld["loss_ctc"] = lc
    ld["loss_total"] = loss
    return loss, ld, logits


def rnnt_greedy(sd, cfg, eouts, elens, prefix="decoder", max_seq_len=256):
    """time-synchronous greedy search, one symbol per joint evaluation (rnn_transducer.py:194-240)"""
    hyps, aligns = [], []
    for b in range(eouts.shape[0]):
        hyp, align = [], []
        ys = torch.full((1, 1), cfg.eos_id, dtype=torch.long)
        dout, state = recurrency(sd, cfg, ys, None, prefix)
        t, T = 0, int(elens[b])
        while t < T:
            tok = int(joint(sd, eouts[b:b + 1, t:t + 1], dout, prefix).squeeze(2).argmax(-1)[0])
            align.append(tok)
            if tok == cfg.blank_id:
                t += 1
            else:
                hyp.append(tok)
                dout, state = recurrency(sd, cfg, torch.tensor([[tok]]), state, prefix)
            if len(hyp) > max_seq_len:
                break
        hyps.append(hyp)
        aligns.append(align)
    return hyps, aligns


def _merge_same_hyp(cands):
    """rnn_transducer.py:348-359: candidates (already sorted best-first) with the same label sequence are
    folded into the first one, whose score becomes the log-sum of the scores; order of first occurrence."""
    seen = {}
    for c in cands:
        key = tuple(c["hyp"])
        if key in seen:
            seen[key]["score"] = float(np.logaddexp(seen[key]["score"], c["score"]))
        else:
            seen[key] = c
    return list(seen.values())


def rnnt_beam_search(sd, cfg, eouts, beam_width, prefix="decoder", num_expands=3):
    """alignment-length synchronous decoding, one utterance (rnn_transducer.py:242-325).

    Every hypothesis carries the LSTM state from BEFORE its last label was consumed (blank extensions keep
    it, :290-294), so each expansion re-runs the prediction network on `hyp[-1]`.  Per frame: up to
    `num_expands` rounds; a round extends every live hypothesis by blank (collected for the next frame) and,
    except in the last round, by its `beam_width` best non-blank labels (:301-312); after each round and
    after each frame the candidates are sorted by score (stable), merged by label sequence and cut to
    `beam_width`.  Returns the surviving label sequences, INCLUDING the leading <sos> (= eos id), best first.
    """
    n, H = cfg.dec_num_layers, cfg.dec_hidden_size
    zero = (torch.zeros(n, 1, H), torch.zeros(n, 1, H))
    beams = [dict(hyp=[cfg.eos_id], score=0.0, state=zero)]
    for t in range(eouts.shape[1]):
        frame_out, live = [], beams
        for v in range(num_expands):
            ys = torch.tensor([[b["hyp"][-1]] for b in live])
            prev = (torch.cat([b["state"][0] for b in live], 1), torch.cat([b["state"][1] for b in live], 1))
            douts, (hs, cs) = recurrency(sd, cfg, ys, prev, prefix)
            lp = torch.log_softmax(joint(sd, eouts[:, t:t + 1], douts, prefix)[:, 0, 0], -1)
            for i, b in enumerate(live):
                frame_out.append(dict(hyp=b["hyp"], score=b["score"] + lp[i, cfg.blank_id].item(), state=b["state"]))
            grown = []
            if v < num_expands - 1:
                for i, b in enumerate(live):
                    top, idx = torch.topk(lp[i, 1:], beam_width)
                    after = (hs[:, i:i + 1], cs[:, i:i + 1])
                    for k in range(beam_width):
                        grown.append(dict(hyp=b["hyp"] + [int(idx[k]) + 1], score=b["score"] + top[k].item(),
                                          state=after))
            grown.sort(key=lambda c: -c["score"])
            live = _merge_same_hyp(grown)[:beam_width]
            if not live:
                break
        frame_out.sort(key=lambda c: -c["score"])
        beams = _merge_same_hyp(frame_out)[:beam_width]
    return [b["hyp"] for b in beams]
