"""Device time per launch of a callable, measured by replaying it from a HIP graph (n launches per replay, best of `reps`):
the figure excludes the Python / ctypes launch path.  Shared by the tools/*_bench.py micro-benchmarks."""
import torch


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
