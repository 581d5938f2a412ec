"""CPU oracle, part 2: Transformer decoder (teacher-forced loss and one-step inference), label
smoothing loss, CTC prefix scorer, joint CTC/attention beam search with LM shallow fusion, and
the Transformer LM's next-token prediction.  TEST INFRASTRUCTURE ONLY (see oracle/model.py).

Citations (file:line in /root/reference):
  decoder layer / stack      asr/modeling/transformer.py:156-198, decoders/transformer.py:82-159
  label smoothing loss       asr/criteria.py:5-46
  joint beam search          asr/modeling/decoders/transformer.py:161-294
  CTC prefix scorer          asr/modeling/decoders/ctc_score.py:13-85
  Transformer LM predict     lm/modeling/transformer.py:62-77, lm/modeling/transformers/modeling_bert.py:159-204,
                             207-303,360-436,516-554, modeling_utils.py:196-245 (causal * padding mask, -10000)
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .model import abs_pos_emb, cfg_get, ctc_loss, ffn, layer_norm, linear, mha, nopad_mask

LOG_0 = -1e10


# ------------------------------------------------------------------ decoder (training)
def label_smoothing_loss(logits, ys, ylens, vocab, lsm_prob, normalize_length=False, normalize_batch=True):
    """-sum_b sum_{t<ylens[b]} sum_v q[b,t,v] * log_softmax(logits)[b,t,v]; q = 1-eps on the label and
    eps/(V-1) elsewhere (criteria.py:9-15); optional /ylen and /B."""
    logp = torch.log_softmax(logits, -1)
    B = logits.shape[0]
    loss = logits.new_zeros(())
    for b in range(B):
        n = int(ylens[b])
        lp = logp[b, :n]
        tgt = lp.gather(1, ys[b, :n].view(-1, 1)).squeeze(1)
        lb = ((1 - lsm_prob) * tgt + (lsm_prob / (vocab - 1)) * (lp.sum(-1) - tgt)).sum()
        if normalize_length:
            lb = lb / n
        loss = loss - lb
    return loss / B if normalize_batch else loss


def tgt_mask(ylens, maxlen=None):
    """key padding AND lower-triangular (model_utils.py:39-43) -> bool [B, L, L]"""
    pad = nopad_mask(ylens, maxlen).unsqueeze(1)
    L = pad.shape[-1]
    return pad & torch.tril(torch.ones(L, L, dtype=torch.bool)).unsqueeze(0)


def decoder_layer(sd, name, h, x, ymask, memory, mmask):
    y = layer_norm(sd, name + ".norm1", x, 1e-12)
    x = x + mha(sd, name + ".self_attn", h, y, y, ymask)
    y = layer_norm(sd, name + ".norm2", x, 1e-12)
    x = x + mha(sd, name + ".src_attn", h, y, memory, mmask)
    return x + ffn(sd, name + ".feed_forward", layer_norm(sd, name + ".norm3", x, 1e-12), F.relu)


def decoder_logits(sd, cfg, eouts, elens, ys_in, ylens_in, prefix="decoder", memory_mask=True):
    """teacher-forced logits [B, L, V]; ylens_in counts <sos> (= ylens + 1 in training)."""
    d, h = cfg.dec_hidden_size, cfg.dec_num_attention_heads
    x = F.embedding(ys_in, sd[prefix + ".embed.weight"]) * math.sqrt(d) + abs_pos_emb(ys_in.shape[1], d)
    ymask = tgt_mask(ylens_in, ys_in.shape[1])
    mmask = nopad_mask(elens, eouts.shape[1]).unsqueeze(1) if memory_mask else None
    for i in range(cfg.dec_num_layers):
        x = decoder_layer(sd, f"{prefix}.transformers.{i}", h, x, ymask, eouts, mmask)
    return linear(sd, prefix + ".output", layer_norm(sd, prefix + ".norm", x, 1e-12))


def decoder_forward(sd, cfg, eouts, elens, ys, ylens, ys_in, ys_out, prefix="decoder"):
    """TransformerDecoder.forward (decoders/transformer.py:82-146): attention loss + mtl_ctc_weight * CTC."""
    logits = decoder_logits(sd, cfg, eouts, elens, ys_in, ylens + 1, prefix)
    loss_att = label_smoothing_loss(logits, ys_out, ylens + 1, cfg.vocab_size, cfg.lsm_prob,
                                    cfg.loss_normalize_length, cfg.loss_normalize_batch)
    loss, loss_dict = loss_att, {"loss_att": loss_att}
    if cfg.mtl_ctc_weight > 0:
        ctc_logits = linear(sd, prefix + ".ctc.output", eouts)
        loss_ctc = ctc_loss(ctc_logits, ys, elens, ylens, cfg.blank_id)
        loss = loss + cfg.mtl_ctc_weight * loss_ctc
        loss_dict["loss_ctc"] = loss_ctc
    loss_dict["loss_total"] = loss
    return loss, loss_dict, logits


def forward_one_step(sd, cfg, ys_in, ylens_in, eouts, prefix="decoder"):
    """logits of the LAST position; the whole prefix is recomputed, cross-attention unmasked
    (decoders/transformer.py:148-159)."""
    logits = decoder_logits(sd, cfg, eouts, None, ys_in, ylens_in, prefix, memory_mask=False)
    return logits[:, -1]


# ------------------------------------------------------------------ CTC prefix scorer
class CTCPrefixScorer:
    """log-probability of all label sequences starting with a prefix (Watanabe et al.); state
    r[t] = (log p(prefix ends in non-blank at t), log p(prefix ends in blank at t))."""

    def __init__(self, x, blank_id, eos_id):
        self.x, self.blank, self.eos, self.T = x, blank_id, eos_id, len(x)

    def initial_state(self):
        r = np.full((self.T, 2), LOG_0, dtype=np.float32)
        r[:, 1] = np.cumsum(self.x[:, self.blank], dtype=np.float32)
        # the reference accumulates in float32 step by step (ctc_score.py:29-31)
        acc = np.float32(self.x[0, self.blank])
        r[0, 1] = acc
        for t in range(1, self.T):
            acc = np.float32(acc + self.x[t, self.blank])
            r[t, 1] = acc
        return r

    def __call__(self, y, cs, r_prev):
        out_len = len(y) - 1
        cs = np.asarray(cs)
        C = len(cs)
        xs = self.x[:, cs]
        r = np.full((self.T, 2, C), LOG_0, dtype=np.float32)  # rows the reference leaves uninitialised are never read
        if out_len == 0:
            r[0, 0] = xs[0]
        r_sum = np.logaddexp(r_prev[:, 0], r_prev[:, 1])
        last = y[-1]
        log_phi = np.repeat(r_sum[:, None], C, axis=1).astype(np.float32)
        if out_len > 0:
            same = cs == last
            log_phi[:, same] = r_prev[:, 1][:, None]
        start = max(out_len, 1)
        log_psi = r[start - 1, 0].copy()
        for t in range(start, self.T):
            r[t, 0] = np.logaddexp(r[t - 1, 0], log_phi[t - 1]) + xs[t]
            r[t, 1] = np.logaddexp(r[t - 1, 0], r[t - 1, 1]) + self.x[t, self.blank]
            log_psi = np.logaddexp(log_psi, log_phi[t - 1] + xs[t])
        log_psi[cs == self.eos] = r_sum[-1]
        log_psi[cs == self.blank] = LOG_0
        return log_psi, np.moveaxis(r, 2, 0)


# ------------------------------------------------------------------ Transformer LM
def gelu(x):
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def lm_logits(sd, cfg, ys, ylens, prefix="lm.transformer"):
    """causal BERT forward -> logits [B, L, V]  (post-LayerNorm layers, GELU, tied output embedding)"""
    h = cfg.num_attention_heads
    B, L = ys.shape
    emb = prefix + ".bert.embeddings"
    x = F.embedding(ys, sd[emb + ".word_embeddings.weight"]) + sd[emb + ".position_embeddings.weight"][:L] \
        + sd[emb + ".token_type_embeddings.weight"][0]
    x = layer_norm(sd, emb + ".LayerNorm", x, 1e-12)
    mask = tgt_mask(ylens, L)  # causal * key padding; masked scores get -10000 added
    for i in range(cfg.num_layers):
        lay = f"{prefix}.bert.encoder.layer.{i}"
        q = linear(sd, lay + ".attention.self.query", x).view(B, L, h, -1).transpose(1, 2)
        k = linear(sd, lay + ".attention.self.key", x).view(B, L, h, -1).transpose(1, 2)
        v = linear(sd, lay + ".attention.self.value", x).view(B, L, h, -1).transpose(1, 2)
        s = q @ k.transpose(-1, -2) / math.sqrt(q.shape[-1]) + (~mask).unsqueeze(1).float() * -10000.0
        ctx = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, L, -1)
        x = layer_norm(sd, lay + ".attention.output.LayerNorm", linear(sd, lay + ".attention.output.dense", ctx) + x, 1e-12)
        y = linear(sd, lay + ".output.dense", gelu(linear(sd, lay + ".intermediate.dense", x)))
        x = layer_norm(sd, lay + ".output.LayerNorm", y + x, 1e-12)
    t = prefix + ".cls.predictions"
    x = layer_norm(sd, t + ".transform.LayerNorm", gelu(linear(sd, t + ".transform.dense", x)), 1e-12)
    return F.linear(x, sd[t + ".decoder.weight"], sd[t + ".bias"])


def lm_predict(sd, cfg, ys, ylens, prefix="lm.transformer"):
    """log-probabilities of the next token after position ylens[b]-1 (lm/modeling/transformer.py:62-77)"""
    logp = torch.log_softmax(lm_logits(sd, cfg, ys, ylens, prefix), -1)
    return torch.stack([logp[b, int(ylens[b]) - 1] for b in range(ys.shape[0])])


# ------------------------------------------------------------------ joint beam search
def joint_beam_search(sd, cfg, eouts, elens, beam_width, len_weight=0.0, lm=None, lm_weight=0.0,
                      decode_ctc_weight=0.0, prefix="decoder", trace=None):
    """TransformerDecoder.decode for one utterance (decoders/transformer.py:161-294), including its
    quirks: (1) `scores` aliases `scores_att`, so the LM term is already inside scores_att when the
    CTC re-scoring adds it a second time; (2) hypotheses ending in <eos> are finished with
    score + len_weight * len(hyp incl. sos/eos) and dropped when empty; (3) the search stops as soon
    as `beam_width` results exist.  `lm` = (lm_sd, lm_cfg) or None.  Returns (hyps, scores)."""
    assert eouts.shape[0] == 1
    V, eos, blank = cfg.vocab_size, cfg.eos_id, cfg.blank_id
    beams = [dict(hyp=[eos], score=0.0, score_ctc=0.0, ctc_state=None)]
    scorer = None
    if decode_ctc_weight > 0:
        ctc_lp = torch.log_softmax(linear(sd, prefix + ".ctc.output", eouts), -1)[0].numpy()
        scorer = CTCPrefixScorer(ctc_lp, blank, eos)
        beams[0]["ctc_state"] = scorer.initial_state()
        cw = min(V, int(beam_width * 1.5))
    results = []
    for i in range(cfg.max_decode_ylen):
        new_beams = []
        for beam in beams:
            ys_in = torch.tensor([beam["hyp"]])
            ylens_in = torch.tensor([i + 1])
            scores_att = torch.log_softmax(forward_one_step(sd, cfg, ys_in, ylens_in, eouts, prefix), -1)
            scores = scores_att
            if lm_weight > 0:
                scores_lm = lm_predict(lm[0], lm[1], ys_in, ylens_in)
                scores = scores + lm_weight * scores_lm[:, :V]
                scores_att = scores  # quirk 1: in-place += on the alias
            if decode_ctc_weight > 0:
                _, v_topb = torch.topk(scores, k=cw, dim=1)
                cands = v_topb[0].numpy()
                scores_ctc, ctc_state = scorer(beam["hyp"], cands, beam["ctc_state"])
                scores = (1 - decode_ctc_weight) * scores_att[:, v_topb[0]] + decode_ctc_weight * torch.from_numpy(
                    scores_ctc - beam["score_ctc"]).unsqueeze(0)
                if lm_weight > 0:
                    scores = scores + lm_weight * scores_lm[:, v_topb[0]]
                scores_topk, ids_topk = torch.topk(scores, k=beam_width, dim=1)
                v_topk = v_topb[:, ids_topk[0]]
            else:
                scores_topk, v_topk = torch.topk(scores, k=beam_width, dim=1)
            if trace is not None:
                trace.append((i, list(beam["hyp"]), scores_topk[0].tolist(), v_topk[0].tolist()))
            for j in range(beam_width):
                nb = dict(score=beam["score"] + float(scores_topk[0, j]), hyp=beam["hyp"] + [int(v_topk[0, j])],
                          score_ctc=0.0, ctc_state=None)
                if decode_ctc_weight > 0:
                    nb["score_ctc"] = scores_ctc[ids_topk[0, j]]
                    nb["ctc_state"] = ctc_state[ids_topk[0, j]]
                new_beams.append(nb)
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
        if len(results) >= beam_width:
            break
        beams = alive
    results = sorted(results, key=lambda r: r["score"], reverse=True)
    return [r["hyp"] for r in results], [r["score"] for r in results]
