"""Critic epoch at the C2 shape (4096 x 500 rows, D_in 40): exact-fp32 kernel vs split-f16 kernel."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd import critic_ops
from tce_rl_amd.nn import MLP
torch.manual_seed(0)
act = sys.argv[1] if len(sys.argv) > 1 else "relu"
din = int(sys.argv[2]) if len(sys.argv) > 2 else 40
mlp = MLP("ValueFunction", din, 1, [128, 128], "orthogonal", 1.0, act, None, torch.float32, torch.device("cuda"))
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(2_000_000); s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize(); best = min(best, s.elapsed_time(e) / n)
    return best
x = torch.randn(4096, 501, din + 8, device="cuda")[:, :-1, :din]
ret = torch.randn(4096, 500, device="cuda")
for arith in ("f32", "f16x2"):
    run = critic_ops.EpochRunner(mlp, arith=arith)
    stats = torch.zeros(2, device="cuda")
    ms = t(lambda: run.epoch(x, ret, ret, 0.0, stats=stats))
    flops = 2.05e6 * 2 * 3 * (din * 128 + 128 * 128 + 128)
    print(f"{act} D_in {din} {arith:6s}: {ms*1e3:8.1f} us/epoch  {flops/ms/1e9:7.1f} TFLOP/s (algorithmic)", flush=True)
