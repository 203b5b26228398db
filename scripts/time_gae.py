import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd import ops
N, T, P = 4096, 500, 24
r = torch.randn(N, T, device="cuda"); v = torch.randn(N, T + 1, device="cuda")
d = torch.zeros(N, T, dtype=torch.bool, device="cuda"); d[:, -1] = True
tl = torch.zeros_like(d)
idx = torch.arange(4, T, 20)
pairs = torch.stack([idx[:-1], idx[1:]], 1).cuda()
for fused in (False, True):
    for _ in range(5): ops.gae(r, v, d, tl, 1.0, 0.95, True, pairs if fused else None)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): ops.gae(r, v, d, tl, 1.0, 0.95, True, pairs if fused else None)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1000 / 50
    b = N * T * 18 + N * 4 + (N * P * 12 if fused else 0)
    print(f"gae fused={fused}: {us:.1f} us/call, {b / us / 1e6:.2f} TB/s algorithmic ({b/us/1e6/8*100:.0f}% of 8 TB/s)")
