"""L2 training step time with dropout 0.1 (bench setting) against dropout 0: what the counter-based masks cost"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
import bench
from emoasr_amd.modeling.asr import ASR
from emoasr_amd.train import ArenaAdam, noam_lr
dev = torch.device("cuda:0")
batches = bench.make_batches(0, 1, 16, dev)
for p in (0.1, 0.0):
    torch.manual_seed(0)
    cfg = dict(bench.L2, dropout_enc_rate=p, dropout_attn_rate=p)
    model = ASR(SimpleNamespace(**cfg), compute_dtype=torch.bfloat16).to(dev).train()
    eng = model.engine()
    opt = ArenaAdam(eng.arena, lambda s: noam_lr(5.0, 256, 25000, s), weight_decay=1e-6, clip_grad_norm=5.0)
    def step(bt):
        loss, _ = model(bt.xs, bt.xlens, bt.ys, bt.ylens, None, None)
        opt.zero_grad(); loss.backward(); opt.step()
    for bt in batches[:4]: step(bt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for bt in batches[4:]: step(bt)
    torch.cuda.synchronize()
    print(f"dropout {p}: {1e3 * (time.perf_counter() - t0) / 12:.3f} ms/step", flush=True)
    del model, eng, opt
