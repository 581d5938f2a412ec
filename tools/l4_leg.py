import sys, json, torch
sys.path.insert(0, '/root/repo')
import bench
from emoasr_amd.hostenv import respect_cpu_quota
respect_cpu_quota()
dev = torch.device('cuda:0')
import os
r = bench.l4_rnnt(dev, torch.bfloat16, steps=int(os.environ.get("L4_STEPS", 4)), n_dec=1)
print(round(r['train_frames_per_s']), round(r['ms_per_step'], 2))
