import cProfile, pstats, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--steps", "6", "--warmup", "3", "--no-cpu-baseline", "--no-decode"]
import torch
import bench
# monkeypatch sync to measure CPU-only time per step
orig_sync = torch.cuda.synchronize
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
st = pstats.Stats(pr, stream=sys.stderr)
st.sort_stats("tottime").print_stats(35)
