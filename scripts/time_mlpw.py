"""Device time of one wide / fp64 critic epoch (chain + gradient + finish) at
the C3 shape.  python scripts/time_mlpw.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd import critic_ops
from tce_rl_amd.nn import MLP
N, T, din, H = 8192, 100, 22, 256
for dt, peak in ((torch.float32, 157.3), (torch.float64, 78.6)):
    torch.manual_seed(0)
    mlp = MLP("ValueFunction", din, 1, [H, H], "orthogonal", 1.0, "leaky_relu", None, dt, torch.device("cuda"))
    x = torch.randn(N, T + 1, 36, device="cuda", dtype=dt)[:, :-1, :din]
    ret = torch.randn(N, T, device="cuda", dtype=dt)
    run = critic_ops.make_runner(mlp)
    for _ in range(2):
        run.epoch(x, ret, ret, 0.0)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        run.epoch(x, ret, ret, 0.0)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    fl = N * T * 2.0 * (3 * (din * H + H * H) - din * H + 3 * H)
    print("%s: %.3f ms / epoch -> %.1f TFLOP/s = %.1f %% of %.1f" % (dt, ms, fl / ms / 1e9, 100 * fl / ms / 1e9 / peak, peak))
