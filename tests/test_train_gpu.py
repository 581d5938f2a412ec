"""The training driver (train.train_step / train, optimizers.Adam + ScheduledOptimizer) against the 12-step
trace the reference's own ScheduledOptimizer + torch.optim.Adam produced on the l2_tiny weights
(tests/golden/train_trace.npz: accum_grad 2, clip 5.0 -- active at every step --, noam warm-up, weight decay).

f32 mode; tolerances: learning rates exact, losses 2e-3 relative (12 steps of error amplification on top of the
1e-3 single-step bar), final parameters 2e-3 of their range, Adam moments 1e-2."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from tests.util import CONFIGS, load_golden

pytestmark = pytest.mark.gpu

TRAIN_TRACE = dict(lr_schedule_type="noam", learning_rate=0.02, num_warmup_steps=4, accum_grad=2, clip_grad_norm=5.0,
                   weight_decay=1e-6, log_step=3)


def _trace():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "train_trace.npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def _data(t, i):
    return {k: t[f"batch{i}/{k}"] for k in ("xs", "xlens", "ys", "ylens", "ys_in", "ys_out")}


def _setup(dev, hip_adam=True):
    from emoasr_amd.modeling.asr import ASR
    from emoasr_amd.optimizers import Adam, ScheduledOptimizer
    _, sd, _ = load_golden("l2_tiny")
    params = SimpleNamespace(**dict(CONFIGS["l2_tiny"], **TRAIN_TRACE))
    model = ASR(params, compute_dtype=torch.float32)
    model.load_state_dict(sd)
    # the reference's order: optimizer first, then model.to(device) (train_asr.py:228-246)
    base = (Adam if hip_adam else torch.optim.Adam)(model.parameters(), lr=0, weight_decay=params.weight_decay)
    optimizer = ScheduledOptimizer(base, params)
    model.to(dev)
    model.train()
    return model, optimizer, params


@pytest.mark.parametrize("hip_adam", [True, False], ids=["hip-adam", "torch-adam"])
def test_twelve_step_trace(dev, hip_adam):
    from emoasr_amd.train import train_step
    t = _trace()
    model, optimizer, params = _setup(dev, hip_adam)
    if not hip_adam:  # torch.optim.Adam must see the arena-backed tensors: bind before its first step
        model.engine()
    optimizer.update_epoch()
    losses, lrs = [], []
    for micro in range(24):
        stepping = (micro + 1) % params.accum_grad == 0
        ld = train_step(model, optimizer, _data(t, micro % 3), params, dev, no_grad=not stepping)
        assert set(ld) == {"loss_ctc", "loss_total"} and isinstance(ld["loss_total"], float)
        losses.append(ld["loss_total"])
        if stepping:
            lrs.append(optimizer._lr)
    assert np.allclose(lrs, t["lrs"].numpy(), rtol=1e-12, atol=0)
    rel = np.abs(np.array(losses) - t["losses"].numpy()) / t["losses"].numpy()
    assert rel.max() < 2e-3, rel
    sd = model.state_dict()
    for k in [k for k in t if k.startswith("end/")]:
        want, got = t[k], sd[k[4:]].cpu()
        if want.dtype.is_floating_point:
            err = (got - want).abs().max().item() / (want.abs().max().item() + 1e-12)
            assert err < 2e-3, (k, err)
        else:
            assert torch.equal(got, want), k
    osd = optimizer.state_dict()
    assert osd["_step"] == int(t["optim/_step"]) and abs(osd["_lr"] - float(t["optim/_lr"])) < 1e-15
    idx = [n for n, _ in model.named_parameters()].index("decoder.output.weight")
    st = osd["optimizer"]["state"][idx]
    assert int(st["step"]) == 12
    for key in ("exp_avg", "exp_avg_sq"):
        want = t[f"optim/{key}/decoder.output.weight"]
        err = (st[key].cpu() - want).abs().max().item() / want.abs().max().item()
        assert err < 1e-2, (key, err)


def test_nan_gradient_skips_the_update_and_resume(dev):
    """train_asr.py:88-89: a NaN gradient norm leaves parameters and moments untouched; then an optimizer
    state_dict round trip through a fresh optimizer continues identically (asr/optimizers.py:99-117)"""
    from emoasr_amd.optimizers import Adam, ScheduledOptimizer
    from emoasr_amd.train import train_step
    t = _trace()
    model, optimizer, params = _setup(dev)
    one = SimpleNamespace(**dict(vars(params), accum_grad=1))
    train_step(model, optimizer, _data(t, 0), one, dev)
    before = {k: v.clone() for k, v in model.state_dict().items() if v.dtype.is_floating_point and "running" not in k}
    # poison the gradient of the next step
    loss, _ = model(**{k: (v.to(dev) if k == "xs" else v) for k, v in _data(t, 1).items()})
    loss.backward()
    model.decoder.output.weight.grad[0, 0] = float("nan")
    optimizer.optimizer.clip_grad_norm = one.clip_grad_norm
    optimizer.step()
    optimizer.zero_grad()
    for k, v in before.items():
        assert torch.equal(model.state_dict()[k], v), k
    assert float(model.engine().arena.grad.abs().max()) == 0.0
    # the skipped update does not advance the schedule position or Adam's step count (the reference never calls
    # optimizer.step() on a NaN norm): both follow at the next host synchronisation point (state_dict / update_epoch)
    assert optimizer._step == 2                      # host counter, not yet corrected
    sd_opt = optimizer.state_dict()
    assert optimizer._step == 1 and sd_opt["_step"] == 1
    assert optimizer.optimizer._core._step == 1 and int(optimizer.optimizer._core.skipped.item()) == 0
    assert {int(v["step"]) for v in sd_opt["optimizer"]["state"].values()} == {1}
    # resume: same model weights, a NEW optimizer restored from the state dict, one more step each
    model2, optimizer2, _ = _setup(dev)
    model2.load_state_dict(model.state_dict())
    optimizer2.load_state_dict(sd_opt)
    a = train_step(model, optimizer, _data(t, 2), one, dev)
    b = train_step(model2, optimizer2, _data(t, 2), one, dev)
    assert abs(a["loss_total"] - b["loss_total"]) < 1e-4 * abs(a["loss_total"])
    assert optimizer2._step == optimizer._step and optimizer2._lr == optimizer._lr
    # (atomically accumulated weight gradients make two runs differ in the last bits)
    for (k, v), (_, w) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.allclose(v.float(), w.float(), rtol=1e-4, atol=1e-5), k


def test_train_epoch_driver(dev):
    from emoasr_amd.train import train
    t = _trace()
    model, optimizer, params = _setup(dev)
    lines = []
    steps = train(model, optimizer, [_data(t, i % 3) for i in range(12)], params, dev, epoch=0, log=lines.append)
    assert steps == 6 and optimizer._epoch == 1 and optimizer._step == 6
    assert len(lines) == 2 and "step =      3 /      6" in lines[0] and "loss_total:" in lines[0]
    # the logged mean over 3 steps x 2 micro-batches equals the reference trace's
    want = t["losses"][:6].sum().item() / 3
    got = float(lines[0].split("loss_total: ")[1].split()[0])
    assert abs(got - want) < 2e-3 * want + 1e-3, (got, want)


def test_partial_optimizer_state_resumes_with_zero_moments(dev):
    """a reference optim.ep{N} written by torch.optim.Adam has no entry for parameters that never received a gradient;
    those resume from zero moments (asr/optimizers.py:110-117 + torch.optim.Adam.load_state_dict semantics)"""
    from emoasr_amd import checkpoint
    from emoasr_amd.train import train_step
    t = _trace()
    model, optimizer, params = _setup(dev)
    one = SimpleNamespace(**dict(vars(params), accum_grad=1))
    train_step(model, optimizer, _data(t, 0), one, dev)
    sd = optimizer.state_dict()
    full = sd["optimizer"]["state"]
    dropped = sorted(full)[-3:]
    sd["optimizer"]["state"] = {k: v for k, v in full.items() if k not in dropped}
    model2, optimizer2, _ = _setup(dev)
    model2.load_state_dict(model.state_dict())
    model2.engine().ensure_bound()              # the arena exists, so the state is applied at once
    optimizer2.load_state_dict(sd)
    core, arena = optimizer2.optimizer._bind(), model2.engine().arena
    for i, n in enumerate(arena.module_order):
        o, v = arena.offsets[n], arena.pviews[n]
        m = core.m[o:o + v.numel()]
        if i in dropped:
            assert float(m.abs().max()) == 0.0, n
        else:
            assert torch.equal(m.view(v.shape).cpu(), full[i]["exp_avg"]), n
    bad = dict(sd["optimizer"]["state"])
    bad[10 ** 6] = full[0]
    sd["optimizer"]["state"] = bad
    with pytest.raises(AssertionError):
        checkpoint.load_optimizer_state_dict(core, {"optimizer": sd["optimizer"], "_step": 1})


def _fit_curve(dev, mode, steps):
    """`steps` optimizer steps (two stacked micro-batches each, dropout 0.1, clip, noam) on one fixed set of 7 utterances"""
    from emoasr_amd.modeling.asr import ASR
    from emoasr_amd.optimizers import Adam, ScheduledOptimizer
    from emoasr_amd.train import train_group
    cfg = dict(input_layer="conv2d", feat_dim=80, num_framestacks=1, encoder_type="conformer", decoder_type="ctc",
               pos_encode_type="rel", enc_hidden_size=256, enc_num_attention_heads=4, enc_num_layers=3,
               enc_intermediate_size=512, dropout_enc_rate=0.1, dropout_attn_rate=0.1, vocab_size=96, blank_id=0, eos_id=2,
               kd_weight=0, lr_schedule_type="noam", learning_rate=0.3, num_warmup_steps=25, accum_grad=2, clip_grad_norm=5.0,
               weight_decay=1e-6, log_step=1000)
    params = SimpleNamespace(**cfg)
    torch.manual_seed(0)
    model = ASR(params, compute_dtype=mode)
    optimizer = ScheduledOptimizer(Adam(model.parameters(), lr=0, weight_decay=params.weight_decay), params)
    model.to(dev).train()
    optimizer.update_epoch()
    g = torch.Generator().manual_seed(4)

    def batch(xlens):
        xlens = torch.tensor(xlens)
        ylens = torch.clamp(xlens // 30, min=1)
        B, T, L = len(xlens), int(xlens.max()), int(ylens.max())
        xs = torch.randn(B, T, 80, generator=g)
        ys = torch.randint(3, 96, (B, L), generator=g)
        for b in range(B):
            xs[b, xlens[b]:] = 0
            ys[b, ylens[b]:] = 2
        return dict(xs=xs, xlens=xlens, ys=ys, ylens=ylens, ys_in=None, ys_out=None)

    datas = [batch([203, 187, 150, 96]), batch([303, 290, 221])]
    curve = []
    for _ in range(steps):
        dicts = train_group(model, optimizer, datas, params, dev)
        curve.append(sum(d["loss_total"] for d in dicts))
    return np.array(curve)


def test_training_converges_alike_in_all_modes(dev):
    """the same 120 optimizer steps (stacked micro-batches, dropout, clipping, warm-up to the reference's peak learning rate scale)
    in f32, f32x3 and bf16 from the same initial weights on one fixed set of utterances (the modes draw the same dropout masks: they
    are a function of seed, step and element index).  Every mode fits the set (CTC loss 208 -> < 1 % of it); while the loss is
    still falling through its first decade (20 steps: 208 -> 18) the curves agree step by step -- f32x3 to 1 %, bf16 to 5 %; each
    mode crosses 5 % of the starting loss within 3 steps of the f32 run (all at step 30).  (From the steep phase on, two runs of the
    SAME mode already differ by 5 % -- float atomics in the weight gradients --, and near zero the runs over-fit along different
    paths: ratios of losses of 0.02-0.3 say nothing.)"""
    steps = 120
    c32 = _fit_curve(dev, torch.float32, steps)
    cx3 = _fit_curve(dev, "f32x3", steps)
    c16 = _fit_curve(dev, torch.bfloat16, steps)
    tail = lambda c: float(c[-20:].mean())
    cross = lambda c: int(np.argmax(c < 0.05 * c[0]))
    early = lambda c: float(np.abs(c[:20] / c32[:20] - 1).max())
    for name, c in (("f32", c32), ("f32x3", cx3), ("bf16", c16)):
        print(f"[measured] {name:6s} loss at steps 1 / 10 / 20 / 40 / 80 / 120: " + " ".join(f"{c[i]:8.2f}" for i in (0, 9, 19, 39, 79, 119))
              + f"; crosses 5 % at step {cross(c) + 1}; first-20 max gap to f32 {early(c):.4f}; last-20 mean {tail(c):.3f}")
    for c in (c32, cx3, c16):
        assert np.isfinite(c).all() and tail(c) < 0.01 * c[0], (c[0], tail(c))
        assert abs(cross(c) - cross(c32)) <= 3, (cross(c), cross(c32))
    assert early(cx3) < 0.01, early(cx3)
    assert early(c16) < 0.05, early(c16)
