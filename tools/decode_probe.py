import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
import bench
from emoasr_amd import ops
from emoasr_amd.modeling.asr import ASR
from emoasr_amd.modeling.lm import LM
from emoasr_amd.decode_rt import DecoderStepRuntime, LMStepRuntime
dev = torch.device("cuda:0")
torch.manual_seed(1)
model = ASR(SimpleNamespace(**bench.L3), compute_dtype=torch.bfloat16).to(dev).eval()
lm = LM(SimpleNamespace(**bench.LM12), compute_dtype=torch.bfloat16).to(dev).eval()
eng = model.engine()
with torch.no_grad():
    eouts, elens, _ = model.encoder(torch.randn(1, 1200, 80, device=dev), [1200])
    rt, lmrt = DecoderStepRuntime(eng), LMStepRuntime(lm)
    rt.begin(eouts, 10)
    for L in (5, 20, 40):
        ys = torch.randint(3, 10000, (10, L))
        def t(fn, n=20):
            fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): fn()
            h = time.perf_counter() - t0; torch.cuda.synchronize()
            return 1e3 * h / n, 1e3 * (time.perf_counter() - t0) / n
        def dec():
            with ops.stream_scope(): rt.step(ys)
        def lms():
            with ops.stream_scope(): lmrt.step(ys)
        print(f"L={L}: decoder step host {t(dec)[0]:.3f} ms, with sync {t(dec)[1]:.3f} ms; LM step host {t(lms)[0]:.3f}, with sync {t(lms)[1]:.3f}")
    # the rest of one beam-search step
    import numpy as np
    from emoasr_amd.engine import h2d_i32
    T, V, nb, cw = eouts.shape[1], 10000, 10, 15
    x = ops.log_softmax(eng.head_logits(eouts, "decoder.ctc.output").view(T, V))
    init_state = ops.ctc_prefix_init(x, 0)
    last = torch.randn(nb, V, device=dev).bfloat16(); lm_lp = torch.randn(nb, V, device=dev)
    def part_a():
        sp = ops.log_softmax(last, add=lm_lp, mu=0.3)
        return ops.topk(sp, cw, aux=lm_lp)
    vals, cands, lm_at = part_a()
    lt = h2d_i32([5] * nb, dev); ol = h2d_i32([3] * nb, dev); pa = h2d_i32(list(range(nb)), dev); pc = h2d_i32([0] * nb, dev)
    _, states = ops.ctc_prefix_score(x, cands, lt, ol, 0, 2, None, pa, pc, init_state)
    def part_b():
        return ops.ctc_prefix_score(x, cands, lt, ol, 0, 2, states, pa, pc, init_state)
    def part_c():
        return [h2d_i32([5] * nb, dev) for _ in range(4)]
    def part_d():
        return vals.cpu().numpy(), cands.cpu().numpy(), lm_at.cpu().numpy()
    for name, fn in (("log_softmax+topk", part_a), ("ctc_prefix_score", part_b), ("4x h2d_i32", part_c), ("3x .cpu()", part_d)):
        h, s = t(fn)
        print(f"{name:18s} host {h:.3f} ms, with sync {s:.3f} ms")
