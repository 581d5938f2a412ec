"""CPU oracle: a plain-PyTorch fp32, functional restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in emoasr_amd/ imports this package; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.

Every function takes the reference's own `state_dict` (same key names / shapes, see
SURVEY.md section 8b) plus a config object with the reference's YAML keys, so the same
weights drive the reference (in the authoring container), this oracle and the HIP engine.
Pinned against the reference by tests/golden/make_golden.py -> tests/golden/*.npz and
tests/test_oracle_golden.py.  Dropout is not modelled (parity runs use p = 0 / eval).

Citations are to /root/reference (emonosuke/emoASR):
  encoder      asr/modeling/encoders/transformer.py:84-113, encoders/conv.py:20-28
  conformer    asr/modeling/conformer.py:16-54 (rel pos table), :57-95 (rel MHSA),
               :98-143 (conv module), :191-229 (layer)
  transformer  asr/modeling/transformer.py:15-45 (abs PE), :48-99 (MHA), :102-118 (FFN),
               :121-153 (encoder layer), :156-198 (decoder layer)
  masks        asr/modeling/model_utils.py:6-43
  CTC          asr/modeling/decoders/ctc.py:87-115 (loss), :176-201 (greedy)
"""
import math
from itertools import groupby

import torch
import torch.nn.functional as F


def cfg_get(cfg, key, default=None):
    return getattr(cfg, key) if hasattr(cfg, key) else default


# ------------------------------------------------------------------ masks / tables
def nopad_mask(lens, maxlen=None):
    """bool [B, maxlen], True where t < lens[b]  (model_utils.py:6-28)"""
    lens = torch.as_tensor(lens)
    maxlen = int(lens.max()) if maxlen is None else maxlen
    return torch.arange(maxlen).unsqueeze(0) < lens.view(-1, 1).cpu()


def sinusoid(positions, d):
    """rows: [sin(p*w0), cos(p*w0), sin(p*w1), ...]  with w_k = 10000^(-2k/d)"""
    positions = torch.as_tensor(positions, dtype=torch.float32).view(-1, 1)
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    out = torch.zeros(positions.shape[0], d)
    out[:, 0::2] = torch.sin(positions * div)
    out[:, 1::2] = torch.cos(positions * div)
    return out


def rel_pos_emb(T, d):
    """[2T-1, d]; row r holds the sinusoid of relative offset rel = T-1-r
    (conformer.py:31-45,50-53: flipped positive part followed by the negative part)."""
    return sinusoid(torch.arange(T - 1, -T, -1), d)


def abs_pos_emb(T, d):
    return sinusoid(torch.arange(T), d)


# ------------------------------------------------------------------ building blocks
def linear(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def layer_norm(sd, name, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def swish(x):
    return x * torch.sigmoid(x)


def ffn(sd, name, x, act):
    return linear(sd, name + ".w2", act(linear(sd, name + ".w1", x)))


def _split_heads(x, h):
    B, T, D = x.shape
    return x.view(B, T, h, D // h).transpose(1, 2)


def _attend(sd, name, v, scores, mask):
    """softmax with key mask (transformer.py:73-94): masked scores -> finfo.min, probabilities
    of masked keys forced to 0 afterwards."""
    if mask is not None:
        dead = ~mask.unsqueeze(1)  # (B,1,*,Tk)
        scores = scores.masked_fill(dead, torch.finfo(scores.dtype).min)
        attn = torch.softmax(scores, -1).masked_fill(dead, 0.0)
    else:
        attn = torch.softmax(scores, -1)
    ctx = attn @ v
    B, h, T, dk = ctx.shape
    return linear(sd, name + ".linear_out", ctx.transpose(1, 2).reshape(B, T, h * dk))


def mha(sd, name, h, query, memory, mask):
    q = _split_heads(linear(sd, name + ".linear_q", query), h)
    k = _split_heads(linear(sd, name + ".linear_k", memory), h)
    v = _split_heads(linear(sd, name + ".linear_v", memory), h)
    scores = q @ k.transpose(-1, -2) / math.sqrt(q.shape[-1])
    return _attend(sd, name, v, scores, mask)


def rel_mha(sd, name, h, x, pos_emb, mask):
    """scores[i,j] = ((q_i+u).k_j + (q_i+v).W_pos.sinusoid(i-j)) / sqrt(dk)  (conformer.py:77-95;
    the rel_shift of :68-75 is the index map r = T-1-(i-j) into the projected table)."""
    q = _split_heads(linear(sd, name + ".linear_q", x), h)
    k = _split_heads(linear(sd, name + ".linear_k", x), h)
    v = _split_heads(linear(sd, name + ".linear_v", x), h)
    T, dk = q.shape[2], q.shape[3]
    p = F.linear(pos_emb, sd[name + ".linear_pos.weight"]).view(-1, h, dk).transpose(0, 1)  # (h,2T-1,dk)
    ac = (q + sd[name + ".pos_bias_u"].unsqueeze(1)) @ k.transpose(-1, -2)
    bd_all = (q + sd[name + ".pos_bias_v"].unsqueeze(1)) @ p.transpose(-1, -2)  # (B,h,T,2T-1)
    i = torch.arange(T).view(-1, 1)
    j = torch.arange(T).view(1, -1)
    bd = torch.gather(bd_all, 3, (T - 1 - (i - j)).expand(bd_all.shape[0], h, T, T))
    return _attend(sd, name, v, (ac + bd) / math.sqrt(dk), mask)


def conv_module(sd, name, x, training, momentum=0.1, eps=1e-5):
    """pointwise(2C) -> GLU -> depthwise k31 -> BatchNorm1d -> Swish -> pointwise (conformer.py:121-143).
    No masking: padded frames take part in the convolution and in the batch statistics."""
    w1 = sd[name + ".pointwise_conv1.weight"].squeeze(-1)
    y = F.glu(F.linear(x, w1, sd[name + ".pointwise_conv1.bias"]), dim=-1)
    wd = sd[name + ".depthwise_conv.weight"]
    y = F.conv1d(y.transpose(1, 2), wd, sd[name + ".depthwise_conv.bias"], padding=(wd.shape[-1] - 1) // 2,
                 groups=wd.shape[0])
    bn = name + ".batch_norm"
    y = F.batch_norm(y, sd[bn + ".running_mean"], sd[bn + ".running_var"], sd[bn + ".weight"], sd[bn + ".bias"],
                     training=training, momentum=momentum, eps=eps)
    if training and bn + ".num_batches_tracked" in sd:
        sd[bn + ".num_batches_tracked"] += 1
    y = swish(y).transpose(1, 2)
    return F.linear(y, sd[name + ".pointwise_conv2.weight"].squeeze(-1), sd[name + ".pointwise_conv2.bias"])


def conformer_layer(sd, name, h, x, mask, pos_emb, training):
    rel = pos_emb is not None
    x = x + 0.5 * ffn(sd, name + ".feed_forward_macaron", layer_norm(sd, name + ".norm_ff_macaron", x, 1e-5), swish)

    def attn_part(x):
        y = layer_norm(sd, name + ".norm_self_attn", x, 1e-5)
        if rel:
            return x + rel_mha(sd, name + ".self_attn", h, y, pos_emb, mask)
        return x + mha(sd, name + ".self_attn", h, y, y, mask)

    def conv_part(x):
        return x + conv_module(sd, name + ".conv", layer_norm(sd, name + ".norm_conv", x, 1e-5), training)

    x = conv_part(attn_part(x)) if rel else attn_part(conv_part(x))  # conformer.py:198-219
    x = x + 0.5 * ffn(sd, name + ".feed_forward", layer_norm(sd, name + ".norm_ff", x, 1e-5), swish)
    return layer_norm(sd, name + ".norm_final", x, 1e-5)


def transformer_enc_layer(sd, name, h, x, mask):
    y = layer_norm(sd, name + ".norm1", x, 1e-12)
    x = x + mha(sd, name + ".self_attn", h, y, y, mask)
    return x + ffn(sd, name + ".feed_forward", layer_norm(sd, name + ".norm2", x, 1e-12), F.relu)


def conv2d_subsample(sd, name, xs, xlens):
    """encoders/conv.py:20-28: two Conv2d(k3,s2)+ReLU over the zero-padded batch, flatten
    channel-major (index c*F2 + f), Linear; lengths ((L-1)//2-1)//2."""
    y = F.relu(F.conv2d(xs.unsqueeze(1), sd[name + ".conv.0.weight"], sd[name + ".conv.0.bias"], stride=2))
    y = F.relu(F.conv2d(y, sd[name + ".conv.2.weight"], sd[name + ".conv.2.bias"], stride=2))
    B, C, T2, F2 = y.shape
    y = linear(sd, name + ".output", y.transpose(1, 2).reshape(B, T2, C * F2))
    return y, ((xlens - 1) // 2 - 1) // 2


# ------------------------------------------------------------------ encoder / CTC
def encoder_forward(sd, cfg, xs, xlens, training=False, prefix="encoder", collect=None):
    """-> (eouts [B,T',d], elens).  `collect`, if a list, receives each layer's output."""
    d, h = cfg.enc_hidden_size, cfg.enc_num_attention_heads
    conformer = cfg.encoder_type == "conformer"
    pos_type = cfg_get(cfg, "pos_encode_type", "abs")
    x, elens = conv2d_subsample(sd, prefix + ".conv", xs, xlens)
    T = x.shape[1]
    mask = nopad_mask(elens, T).unsqueeze(1)  # (B,1,T') key mask
    if collect is not None:
        collect.append(x)
    if pos_type == "rel":
        x = x * math.sqrt(d)
        pos_emb = rel_pos_emb(T, d)
    else:
        x = x * math.sqrt(d) + abs_pos_emb(T, d)
        pos_emb = None
    for i in range(cfg.enc_num_layers):
        name = f"{prefix}.transformers.{i}"
        if conformer:
            x = conformer_layer(sd, name, h, x, mask, pos_emb, training)
        else:
            x = transformer_enc_layer(sd, name, h, x, mask)
        if collect is not None:
            collect.append(x)
    return layer_norm(sd, prefix + ".norm", x, 1e-12), elens


def ctc_loss(logits, ys, elens, ylens, blank):
    """sum over utterances of -log p(y|x), zero_infinity, divided by B (ctc.py:36-38,109-113)."""
    lp = logits.transpose(0, 1).log_softmax(2)
    return F.ctc_loss(lp, ys, elens, ylens, blank=blank, reduction="sum", zero_infinity=True) / logits.shape[0]


def ctc_decoder_forward(sd, cfg, eouts, elens, ys=None, ylens=None, prefix="decoder"):
    logits = linear(sd, prefix + ".output", eouts)
    if ys is None:
        return logits
    loss = ctc_loss(logits, ys, elens, ylens, cfg.blank_id)
    return loss, {"loss_ctc": loss, "loss_total": loss}, logits


def ctc_greedy(logits, elens, blank):
    """argmax of raw logits, collapse repeats, drop blanks; <eos> is kept (ctc.py:176-201)."""
    best = logits.argmax(-1)
    hyps, aligns = [], []
    for b in range(logits.shape[0]):
        idx = best[b, : int(elens[b])].tolist()
        aligns.append(idx)
        hyps.append([k for k, _ in groupby(idx) if k != blank])
    return hyps, aligns


def asr_ctc_forward(sd, cfg, xs, xlens, ys, ylens, training=False):
    """ASR.forward for decoder_type == 'ctc' (asr.py:53-68): trims padding to the batch maxima."""
    xs = xs[:, : int(xlens.max())]
    ys = ys[:, : int(ylens.max())]
    eouts, elens = encoder_forward(sd, cfg, xs, xlens, training)
    return ctc_decoder_forward(sd, cfg, eouts, elens, ys, ylens)


def asr_ctc_greedy(sd, cfg, xs, xlens):
    eouts, elens = encoder_forward(sd, cfg, xs, xlens, False)
    logits = ctc_decoder_forward(sd, cfg, eouts, elens)
    hyps, aligns = ctc_greedy(logits, elens, cfg.blank_id)
    return hyps, aligns, logits
