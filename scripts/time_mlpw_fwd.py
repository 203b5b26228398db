"""Device time of the forward-only wide critic (rollout values) at the C3 shape."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd import critic_ops
from tce_rl_amd.nn import MLP
N, T, din, H = 8192, 100, 22, 256
for dt in (torch.float32, torch.float64):
    torch.manual_seed(0)
    mlp = MLP("ValueFunction", din, 1, [H, H], "orthogonal", 1.0, "leaky_relu", None, dt, torch.device("cuda"))
    x = torch.randn(N, T + 1, 36, device="cuda", dtype=dt)[..., :din]
    for _ in range(3):
        critic_ops.forward(mlp, x)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(4):
        s.record()
        for _ in range(10):
            critic_ops.forward(mlp, x)
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 10)
    fl = N * (T + 1) * 2.0 * (din * H + H * H + H)
    print("%s forward: %.3f ms -> %.1f TFLOP/s" % (dt, best, fl / best / 1e9))
