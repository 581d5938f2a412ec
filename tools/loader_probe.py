"""the RTF fixture's loader alone on the GPU box's host: ms per item, cold and warm, and with the GPU decode interleaved"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
with tempfile.TemporaryDirectory() as td:
    print("tmpdir", td, flush=True)
    loader, vocab, _ = bench.rtf_fixture(os.path.join(td, "g"), 20, 1)
    for rep in range(3):
        t0 = time.perf_counter(); n = 0
        for d in loader: n += 1
        print(f"loader only, rep {rep}: {1e3 * (time.perf_counter() - t0) / n:.2f} ms per item", flush=True)
    fp = os.path.join(td, "g", "utt0.npy")
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(20): a = np.load(fp)
        print(f"np.load x20: {1e3 * (time.perf_counter() - t0) / 20:.2f} ms each ({a.nbytes / 1e3:.0f} KB)", flush=True)
    import torch
    x = torch.randn(1000, 1000, device="cuda:0")
    for rep in range(3):
        t0 = time.perf_counter(); n = 0
        for d in loader:
            n += 1
            y = x @ x
            torch.cuda.synchronize()
        print(f"loader + a GPU sync per item, rep {rep}: {1e3 * (time.perf_counter() - t0) / n:.2f} ms per item", flush=True)
