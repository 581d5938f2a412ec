"""Where an utterance of the config-4 decode spends its time outside the search loop."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import logging
import torch
import bench
from emoasr_amd import decode as dec
from emoasr_amd.modeling.asr import ASR
from emoasr_amd.modeling.lm import LM
logging.disable(logging.WARNING)
dev = torch.device("cuda:0")
torch.manual_seed(1)
model = ASR(SimpleNamespace(**bench.L3), compute_dtype=torch.bfloat16).to(dev).eval()
lm = LM(SimpleNamespace(**bench.LM12), compute_dtype=torch.bfloat16).to(dev).eval()
model.decoder.max_decode_ylen = 36
with tempfile.TemporaryDirectory() as td:
    loader, vocab, _ = bench.rtf_fixture(os.path.join(td, "l33"), 6, 2)
    datas = list(loader)
    for rep in range(2):
        for data in datas[:4]:
            torch.cuda.synchronize(); t0 = time.perf_counter()
            xs = data["xs"].to(dev)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            with torch.no_grad():
                eouts, elens, _ = model.encoder(xs, data["xlens"])
            torch.cuda.synchronize(); t2 = time.perf_counter()
            hyps = model.decoder.decode(eouts, elens, None, 10, 0.0, lm, 0.3, 0.3)
            torch.cuda.synchronize(); t3 = time.perf_counter()
            if rep:
                print(f"T {int(data['xlens'][0])}: H2D {1e3*(t1-t0):.2f} ms, encoder {1e3*(t2-t1):.2f} ms, search {1e3*(t3-t2):.2f} ms", flush=True)
    t0 = time.perf_counter()
    n = 0
    for data in loader:
        n += 1
    print(f"loader alone: {1e3 * (time.perf_counter() - t0) / n:.2f} ms per utterance")
