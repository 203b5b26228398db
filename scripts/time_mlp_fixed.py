"""Fixed (per-launch) cost of the fused critic epoch kernel: time vs tiles per workgroup."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd import critic_ops
from tce_rl_amd.nn import MLP
torch.manual_seed(0)
mlp = MLP("ValueFunction", 40, 1, [128, 128], "orthogonal", 1.0, "relu", None, torch.float32, torch.device("cuda"))
run = critic_ops.EpochRunner(mlp)
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(2_000_000); s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize(); best = min(best, s.elapsed_time(e) / n)
    return best
for tiles_per_wg in (1, 2, 4, 8, 16, 125):
    R = 64 * 256 * tiles_per_wg
    x = torch.randn(R // 512, 512, 48, device="cuda")[..., :40]
    ret = torch.randn(R // 512, 512, device="cuda")
    stats = torch.zeros(2, device="cuda")
    ms = t(lambda: run.epoch(x, ret, ret, 0.0, stats=stats))
    msf = t(lambda: critic_ops.forward(mlp, x))
    print(f"{tiles_per_wg:4d} tiles/WG ({R} rows): bwd epoch {ms*1e3:8.1f} us   fwd {msf*1e3:8.1f} us", flush=True)
