"""What a pure streaming WRITE / COPY of the feed-forward products' output sizes takes on this box (torch fill_ / copy_): the floor
of any kernel that writes 72 / 144 MB, whatever it computes."""
import torch
dev = torch.device("cuda:0")
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mb in (18, 36, 72, 144, 288):
    n = mb * 1000 * 1000 // 2
    a = torch.empty(n, device=dev, dtype=torch.bfloat16); b = torch.empty_like(a)
    w = t(lambda: a.fill_(1.0)); c = t(lambda: b.copy_(a))
    print(f"{mb:4d} MB: fill {w:6.1f} us ({mb / w * 1e-3 * 1e3:5.2f} TB/s written)   copy {c:6.1f} us ({2 * mb / c:5.2f} TB/s read + written)")
