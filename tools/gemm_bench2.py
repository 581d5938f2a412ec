"""Kernel-time micro-benchmark of the NT / NN GEMM entry points on the L2 model's shapes.  The calls are captured
into a HIP graph (20 launches per replay), so the figure is device time per launch, not the Python/ctypes
launch path (tools/gemm_bench.py measures ~10 us per call for anything shorter than that).
Usage: python tools/gemm_bench2.py "gemm_tile=0" "gemm_tile=3,gemm_kb=2" blas ...   (one column per option set;
"blas" = torch.mm, the vendor library, as a reference point only)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from emoasr_amd import lib, ops

dev = torch.device("cuda:0")
M = int(os.environ.get("M", 7029))
dt = torch.bfloat16
shapes = [("nt", "ffn1", M, 1024, 256), ("nt", "ffn2", M, 256, 1024), ("nt", "qkv", M, 768, 256), ("nt", "out", M, 256, 256),
          ("nt", "pw1", M, 512, 256), ("nt", "head", M, 10000, 256), ("nt", "lin", M, 256, 4864),
          ("nn", "d_ffn2", M, 1024, 256), ("nn", "d_ffn1", M, 256, 1024), ("nn", "d_qkv", M, 256, 768),
          ("nn", "d_out", M, 256, 256), ("nn", "d_pw1", M, 256, 512), ("nn", "d_lin", M, 4864, 256),
          ("nn", "dcol", 136800, 2304, 256), ("nt", "big", 136800, 2304, 256)]
configs = sys.argv[1:] or ["gemm_tile=0", "blas"]
DEFAULTS = {"gemm_tile": 0, "gemm_kb": 0, "gemm_xcd": 1}


def set_cfg(c):
    opts = dict(DEFAULTS)
    for kv in c.split(","):
        k, v = kv.split("=")
        opts[k] = int(v)
    for k, v in opts.items():
        lib.set_option(k, v)


def rnd(*s):
    return torch.randn(*s, device=dev).to(dt)


def graph_time(fn, n=20, reps=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        st.synchronize()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


if __name__ == "__main__":
    print(f"{'shape':28s}" + "".join(f"{c:>34s}" for c in configs))
    tot = [0.0] * len(configs)
    for kind, name, m, n, k in shapes:
        a = rnd(m, k)
        b = rnd(n, k) if kind == "nt" else rnd(k, n)
        out = torch.empty(m, n, device=dev, dtype=dt)
        bias = torch.randn(n, device=dev)
        row = f"{kind} {name:7s}{m:7d}x{n:5d}x{k:5d}"
        for ci, c in enumerate(configs):
            if c == "blas":  # reference point only: the vendor BLAS behind torch.mm (not on the product path)
                bb = b.t() if kind == "nt" else b
                us = graph_time(lambda: torch.mm(a, bb, out=out))
                tot[ci] += us
                row += f"{us:18.1f} us {2 * m * n * k / us / 1e6:7.0f} TF/s"
                continue
            set_cfg(c)
            if kind == "nt":
                us = graph_time(lambda: ops.gemm_nt(a, b, out=out, bias=bias))
            else:
                us = graph_time(lambda: ops.gemm_nn(a, b, out=out))
            tot[ci] += us
            row += f"{us:18.1f} us {2 * m * n * k / us / 1e6:7.0f} TF/s"
        print(row, flush=True)
    print(f"{'sum':28s}" + "".join(f"{t:18.1f} us            " for t in tot))
