"""emoasr_beam_scores_topk alone: time against k (rounds) and V (row length): python tools/topk_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emoasr_amd import lib, ops

dev = torch.device("cuda:0")
M = 10
for V in (10000, 2000):
    dec = torch.randn(M, V, device=dev).to(torch.bfloat16)
    lm = torch.randn(M, V, device=dev)
    for k in (1, 15, 30):
        for use_lm in (1, 0):
            vals = torch.empty(M, k, device=dev); idx = torch.empty(M, k, device=dev, dtype=torch.int32); at = torch.empty(M, k, device=dev)
            def f():
                lib.call("emoasr_beam_scores_topk", ops.dt(dec), M, V, k, dec.data_ptr(), V, lm.data_ptr() if use_lm else None, V, 0.3,
                         vals.data_ptr(), idx.data_ptr(), at.data_ptr(), ops._stream())
            for _ in range(5): f()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(100): f()
            e.record(); torch.cuda.synchronize()
            print(f"V {V:6d} k {k:3d} lm {use_lm}: {s.elapsed_time(e) * 10:.1f} us", flush=True)
