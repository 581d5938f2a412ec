cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_26; mkdir -p $O
python3 tools/f32_leg.py --split --stacked 2>&1 | grep -v amdgpu.ids | tee $O/leg.txt
python3 - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a $O/leg.txt
import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import bench
from types import SimpleNamespace
from emoasr_amd.hostenv import respect_cpu_quota
respect_cpu_quota()
dev = torch.device("cuda:0")
batches = bench.make_batches(0, 1, 40, dev)
from emoasr_amd import ops
orig = ops.set_f32_split
def traced(on):
    import traceback
    print("set_f32_split", on, "from", traceback.extract_stack()[-2][2], flush=True)
    return orig(on)
ops.set_f32_split = traced
r = bench.parity_mode(dev, batches)
print({k: v for k, v in r.items() if "frames" in k or "ms_per" in k})
PY
