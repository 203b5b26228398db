import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tce_rl_amd import critic_ops
from tce_rl_amd.nn import MLP
N, T, din, H = 8192, 100, 22, 256
dt = torch.float64 if len(sys.argv) < 2 or sys.argv[1] == "f64" else torch.float32
mlp = MLP("ValueFunction", din, 1, [H, H], "orthogonal", 1.0, "leaky_relu", None, dt, torch.device("cuda"))
x = torch.randn(N, T + 1, 36, device="cuda", dtype=dt)[:, :-1, :din]
ret = torch.randn(N, T, device="cuda", dtype=dt)
run = critic_ops.make_runner(mlp)
for _ in range(6):
    run.epoch(x, ret, ret, 0.0)
torch.cuda.synchronize()
