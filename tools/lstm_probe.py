import sys, torch
sys.path.insert(0, '/root/repo')
from emoasr_amd import ops
dev = torch.device('cuda:0'); dt = torch.bfloat16
H, U = 512, 100
for B in (36, 64, 65, 128, 139, 192, 256, 512):
    pre = (torch.randn(U, B, 4 * H, device=dev) * 0.1).to(dt)
    w_hh = (torch.randn(4 * H, H, device=dev) * 0.04).to(dt)
    hseq = torch.empty(U, B, H, device=dev, dtype=dt); cseq = torch.empty(U, B, H, device=dev, dtype=torch.float32)
    gact = torch.empty(U, B, 4 * H, device=dev, dtype=dt)
    assert ops.lstm_seq_supported(pre, B, H)
    def f(): ops.lstm_seq_fwd(pre, w_hh, None, None, hseq, cseq, gact)
    dh = (torch.randn(U, B, H, device=dev) * 0.1).to(dt); dgp = torch.empty(U, B, 4 * H, device=dev, dtype=dt)
    def g(): ops.lstm_seq_bwd(dh, gact, cseq, None, w_hh, dgp)
    out = []
    for fn in (f, g):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 5 * 1e3)
    print(f"B {B:3d}: fwd {out[0]:7.1f} us ({out[0]/U:.2f} us/pos)  bwd {out[1]:7.1f} us ({out[1]/U:.2f} us/pos)")
