"""where the greedy-decode RTF protocol's time goes (bench.decode_rtf): feature file load, H2D, model.decode, text"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
import bench
from emoasr_amd.hostenv import respect_cpu_quota
if __import__('os').environ.get('EMOASR_NO_QUOTA') != '1':
    respect_cpu_quota()
from emoasr_amd.modeling.asr import ASR

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ASR(SimpleNamespace(**bench.L2), compute_dtype=torch.bfloat16).to(dev)
model.eval()
with tempfile.TemporaryDirectory() as td:
    loader, vocab, _ = bench.rtf_fixture(os.path.join(td, "g"), 20, 1)
    for rep in range(3):
        t_load = t_h2d = t_dec = t_sync = 0.0
        frames = 0
        torch.cuda.synchronize()
        t_all = time.perf_counter()
        it = iter(loader)
        while True:
            t0 = time.perf_counter()
            try:
                d = next(it)
            except StopIteration:
                break
            t1 = time.perf_counter()
            x = d["xs"].to(dev)
            t2 = time.perf_counter()
            hyps, scores, _, _ = model.decode(x, d["xlens"], 1, 0.0, lm=None, lm_weight=0.0, decode_ctc_weight=0.0, decode_phone=False)
            t3 = time.perf_counter()
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            t_load += t1 - t0; t_h2d += t2 - t1; t_dec += t3 - t2; t_sync += t4 - t3
            frames += int(d["xlens"][0])
        tot = time.perf_counter() - t_all
        print(f"rep {rep}: 20 utts, {frames} frames: total {1e3 * tot:.1f} ms = RTF {tot / (frames * 0.01):.2e}; per utt: load {1e3 * t_load / 20:.2f} "
              f"h2d {1e3 * t_h2d / 20:.2f} decode(call) {1e3 * t_dec / 20:.2f} trailing sync {1e3 * t_sync / 20:.2f} ms", flush=True)
