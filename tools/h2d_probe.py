import sys, torch
mode = sys.argv[1]
dev = torch.device("cuda:0")
x = torch.zeros(10, device=dev); torch.cuda.synchronize()
pinned = torch.empty(4096, dtype=torch.int32).pin_memory()
dst = torch.empty(4096, dtype=torch.int32, device=dev)
vals = list(range(50))
for i in range(20):
    if mode == "a":
        t = torch.as_tensor(vals, dtype=torch.int32).pin_memory().to(dev, non_blocking=True)
    elif mode == "b":
        pinned[:50] = torch.as_tensor(vals, dtype=torch.int32)
        dst[:50].copy_(pinned[:50], non_blocking=True)
    elif mode == "c":
        t = torch.tensor(vals, dtype=torch.int32, device=dev)
    x.add_(1.0)
torch.cuda.synchronize()
