"""run-to-run difference of the gradient arena for two identical training steps (bf16, dropout 0.1), per parameter"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_fullsize_gpu import _model, _bench_batch
dev = torch.device("cuda:0")
model = _model(torch.bfloat16, dev, dropout_enc_rate=0.1, dropout_attn_rate=0.1)
model.train()
eng = model.engine()
xs, xlens, ys, ylens = _bench_batch()
xs = xs.to(dev)
def grads():
    eng.step_count = 7
    model.zero_grad(set_to_none=False)
    loss, _ = model(xs, xlens, ys, ylens, None, None)
    loss.backward()
    return eng.arena.grad.clone()
g1, g2 = grads(), grads()
A = eng.arena
rows = []
for n in A.names:
    o = A.offsets[n]; k = A.pviews[n].numel()
    a, b = g1[o:o + k], g2[o:o + k]
    d = (a - b).abs().max().item(); m = a.abs().max().item()
    rows.append((d, m, n))
rows.sort(reverse=True)
print("global max |g|", g1.abs().max().item())
for d, m, n in rows[:25]:
    print(f"{d:10.4e}  max {m:10.4e}  rel {d / (m + 1e-30):8.2e}  {n}")
