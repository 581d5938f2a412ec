"""The reference's d = 256 optimizer trace (tests/golden/train_trace_d256.npz) replayed in three modes: bf16 stacked (train_group),
bf16 one micro-batch after the other, f32 one by one -- update-vector cosines of the stored tensors against the reference's.
Measured: f32 1.0000 everywhere (loss 2.7e-6); both bf16 paths 0.999+ except linear_pos.weight 0.82-0.85 (tests/test_stacked_oracle_gpu.py)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from types import SimpleNamespace
from tests.test_stacked_oracle_gpu import TRACE_CFG, _cos
from tests.util import synthetic_state
from emoasr_amd.modeling.asr import ASR
from emoasr_amd.optimizers import Adam, ScheduledOptimizer
from emoasr_amd.train import train_group, train_step
dev = torch.device("cuda:0")
z = np.load("tests/golden/train_trace_d256.npz"); t = {k: torch.from_numpy(z[k]) for k in z.files}
data = lambda i: {k: t[f"batch{i}/{k}"] for k in ("xs", "xlens", "ys", "ylens", "ys_in", "ys_out")}
for mode in ("bf16-stacked", "bf16-onebyone", "f32-onebyone"):
    params = SimpleNamespace(**TRACE_CFG)
    dt = torch.float32 if mode.startswith("f32") else torch.bfloat16
    model = ASR(params, compute_dtype=dt)
    sd0 = synthetic_state({k: v.shape for k, v in model.state_dict().items()})
    model.load_state_dict(sd0)
    opt = ScheduledOptimizer(Adam(model.parameters(), lr=0, weight_decay=params.weight_decay), params)
    model.to(dev).train(); opt.update_epoch()
    losses = []
    for step in range(12):
        pair = [data((2 * step) % 3), data((2 * step + 1) % 3)]
        if mode == "bf16-stacked":
            losses += [d["loss_total"] for d in train_group(model, opt, pair, params, dev)]
        else:
            for k, d in enumerate(pair):
                losses.append(train_step(model, opt, d, params, dev, no_grad=(k == 0))["loss_total"])
    rel = np.abs(np.array(losses) - t["losses"].numpy()) / t["losses"].numpy()
    sd = model.state_dict()
    out = [f"{mode}: loss rel max {rel.max():.2e}"]
    for k in [k for k in t if k.startswith("end/") and t[k].dtype.is_floating_point and "running" not in k]:
        got, want, init = sd[k[4:]].cpu(), t[k], sd0[k[4:]]
        out.append(f"{k.split('.')[-2][-10:]}.{k.split('.')[-1][:1]} {_cos(got - init, want - init):.4f}")
    print("  ".join(out), flush=True)
