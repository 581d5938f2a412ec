"""Would an f32 RESIDUAL STREAM inside the bf16 engine meet north_star's logits tolerance?  (review item: "a throughput mode that
meets 1e-3".)  CPU simulation on the full-size L2 model with the oracle (oracle/model.py): every tensor the HIP engine holds in
bf16 is rounded to bf16 at the point where the engine rounds it (GEMM operands, LayerNorm outputs, q/k/v, soft-max probabilities,
attention output, the convolution module's intermediates, the FFN activation); accumulation stays f32.  Two variants: the
residual stream rounded to bf16 after every sub-layer (the engine today) or kept in f32 (residual adds and the layers' final
LayerNorm outputs un-rounded).  Measured (python tools/sim_residual_f32.py, 3 utterances of 298-403 frames, random-init weights):
    bf16 residual   logits_rel 1.22e-02   frame arg-max agreement 0.970      (the HIP engine on the same batch: 1.25e-2 / 0.957)
    f32 residual    logits_rel 4.56e-03   frame arg-max agreement 0.985
The f32 residual stream removes under two thirds of the error; what is left is the rounding of the GEMM / attention OPERANDS to
bf16, which any bf16-MFMA engine has.  It does not reach 2e-3, so the mode was not built: the mode that meets 1e-3 stays the f32
engine (compute_dtype=torch.float32).  A split-bf16 (hi + lo, three MFMAs per product) f32-storage mode would; not built."""
import sys, math, torch, torch.nn.functional as F
sys.path.insert(0,'/root/repo')
from types import SimpleNamespace
from oracle import model as om
from tests.test_fullsize_gpu import L2, _batch
from emoasr_amd.modeling.asr import ASR

r = lambda x: x.bfloat16().float()
MODE = {"resid_f32": False, "on": False}

def linear(sd, name, x):
    if not MODE["on"]: return F.linear(x, sd[name+".weight"], sd.get(name+".bias"))
    return F.linear(r(x), r(sd[name+".weight"]), sd.get(name+".bias"))   # f32 accumulate; caller rounds
def layer_norm(sd, name, x, eps):
    y = F.layer_norm(x, (x.shape[-1],), sd[name+".weight"], sd[name+".bias"], eps)
    if not MODE["on"]: return y
    if name.endswith("norm_final") and MODE["resid_f32"]: return y
    return r(y)
def ffn(sd, name, x, act):
    a = act(linear(sd, name+".w1", x))
    if MODE["on"]: a = r(a)
    return linear(sd, name+".w2", a)
om.linear, om.layer_norm, om.ffn = linear, layer_norm, ffn
orig_attend = om._attend
def _attend(sd, name, v, scores, mask):
    if mask is not None:
        dead = ~mask.unsqueeze(1)
        scores = scores.masked_fill(dead, torch.finfo(scores.dtype).min)
        attn = torch.softmax(scores, -1).masked_fill(dead, 0.0)
    else:
        attn = torch.softmax(scores, -1)
    if MODE["on"]: attn, v = r(attn), r(v)
    ctx = attn @ v
    if MODE["on"]: ctx = r(ctx)
    B, h, T, dk = ctx.shape
    return linear(sd, name + ".linear_out", ctx.transpose(1, 2).reshape(B, T, h * dk))
om._attend = _attend
def rel_mha(sd, name, h, x, pos_emb, mask):
    rr = r if MODE["on"] else (lambda t: t)
    q = om._split_heads(rr(linear(sd, name + ".linear_q", x)), h)
    k = om._split_heads(rr(linear(sd, name + ".linear_k", x)), h)
    v = om._split_heads(rr(linear(sd, name + ".linear_v", x)), h)
    T, dk = q.shape[2], q.shape[3]
    p = rr(F.linear(rr(pos_emb), rr(sd[name + ".linear_pos.weight"]))).view(-1, h, dk).transpose(0, 1)
    ac = rr(q + sd[name + ".pos_bias_u"].unsqueeze(1)) @ k.transpose(-1, -2)
    bd_all = rr(q + sd[name + ".pos_bias_v"].unsqueeze(1)) @ p.transpose(-1, -2)
    i = torch.arange(T).view(-1, 1); j = torch.arange(T).view(1, -1)
    bd = torch.gather(bd_all, 3, (T - 1 - (i - j)).expand(bd_all.shape[0], h, T, T))
    return _attend(sd, name, v, (ac + bd) / math.sqrt(dk), mask)
om.rel_mha = rel_mha
def conv_module(sd, name, x, training, momentum=0.1, eps=1e-5):
    rr = r if MODE["on"] else (lambda t: t)
    w1 = sd[name + ".pointwise_conv1.weight"].squeeze(-1)
    y = rr(F.glu(rr(F.linear(rr(x), rr(w1), sd[name + ".pointwise_conv1.bias"])), dim=-1))
    wd = sd[name + ".depthwise_conv.weight"]
    y = rr(F.conv1d(y.transpose(1, 2), wd, sd[name + ".depthwise_conv.bias"], padding=(wd.shape[-1] - 1) // 2, groups=wd.shape[0]))
    bn = name + ".batch_norm"
    y = F.batch_norm(y, sd[bn + ".running_mean"], sd[bn + ".running_var"], sd[bn + ".weight"], sd[bn + ".bias"], training=False, momentum=momentum, eps=eps)
    y = rr(om.swish(y)).transpose(1, 2)
    return F.linear(y, rr(sd[name + ".pointwise_conv2.weight"].squeeze(-1)), sd[name + ".pointwise_conv2.bias"])
om.conv_module = conv_module
def conformer_layer(sd, name, h, x, mask, pos_emb, training):
    rx = (lambda t: t) if (not MODE["on"] or MODE["resid_f32"]) else r
    x = rx(x + 0.5 * ffn(sd, name + ".feed_forward_macaron", layer_norm(sd, name + ".norm_ff_macaron", x, 1e-5), om.swish))
    x = rx(x + rel_mha(sd, name + ".self_attn", h, layer_norm(sd, name + ".norm_self_attn", x, 1e-5), pos_emb, mask))
    x = rx(x + conv_module(sd, name + ".conv", layer_norm(sd, name + ".norm_conv", x, 1e-5), training))
    x = rx(x + 0.5 * ffn(sd, name + ".feed_forward", layer_norm(sd, name + ".norm_ff", x, 1e-5), om.swish))
    return layer_norm(sd, name + ".norm_final", x, 1e-5)
om.conformer_layer = conformer_layer

torch.manual_seed(0)
model = ASR(SimpleNamespace(**L2), compute_dtype=torch.bfloat16)
with torch.no_grad():
    model.decoder.output.weight.mul_(3.0)
    for n, p in model.named_parameters():
        if "batch_norm" in n or ".norm" in n: p.add_(0.05 * torch.randn_like(p))
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
cfg = SimpleNamespace(**L2)
xs, xlens, ys, ylens = _batch(1, [403, 367, 298])
def run():
    with torch.no_grad():
        eouts, elens = om.encoder_forward(sd, cfg, xs, xlens)
        if MODE["on"]: eouts = r(eouts)
        logits = F.linear(r(eouts) if MODE["on"] else eouts, r(sd["decoder.output.weight"]) if MODE["on"] else sd["decoder.output.weight"], sd["decoder.output.bias"])
        hyps, _ = om.ctc_greedy(logits, elens, 0)
    return logits, hyps, elens
ref, h0, elens = run()
for resid in (False, True):
    MODE["on"], MODE["resid_f32"] = True, resid
    lg, h1, _ = run()
    rel = ((lg - ref).abs().max() / ref.abs().max()).item()
    agree = sum(int(a == b) for x, y in zip(h1, h0) for a, b in zip(x, y)) / max(1, sum(len(y) for y in h0))
    fr = sum((lg[b,:int(elens[b])].argmax(-1) == ref[b,:int(elens[b])].argmax(-1)).sum().item() for b in range(3)) / int(sum(elens))
    print("resid_f32" if resid else "bf16 resid", "logits_rel %.2e" % rel, "hyp agree %.3f frame agree %.4f" % (agree, fr), h1 == h0)
