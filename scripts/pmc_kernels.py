"""Launch the roofline kernels a few times (for rocprofv3 --pmc passes)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd import ops, critic_ops
from tce_rl_amd.nn import MLP
from tce_rl_amd.mp import ProDMP
N, T = 4096, 500
g = torch.Generator(device="cuda").manual_seed(0)
r = torch.randn(N, T, device="cuda", generator=g); v = torch.randn(N, T + 1, device="cuda", generator=g)
d = torch.zeros(N, T, dtype=torch.bool, device="cuda"); d[:, -1] = True; tl = torch.zeros_like(d)
big = torch.empty(1 << 28, device="cuda")           # 1 GiB flush buffer (> 256 MiB MALL)
for _ in range(3):
    big.fill_(1.0)
    ops.gae(r, v, d, tl, 1.0, 0.95, True)
torch.manual_seed(0)
mlp = MLP("ValueFunction", 40, 1, [128, 128], "orthogonal", 1.0, "relu", None, torch.float32, torch.device("cuda"))
full = torch.randn(N, T + 1, 48, device="cuda", generator=g); x = full[:, :-1, :40]
run = critic_ops.EpochRunner(mlp)
for _ in range(3):
    big.fill_(1.0)
    run.epoch(x, r, r, 0.0)
run16 = critic_ops.EpochRunner(mlp, arith="f16x2")
for _ in range(3):
    big.fill_(1.0)
    run16.epoch(x, r, r, 0.0)
for _ in range(3):
    big.fill_(1.0)
    critic_ops.forward(mlp, x)
mp = ProDMP(dtype=torch.float32, device="cuda", num_dof=4, num_basis=5, tau=5, alpha_phase=3, alpha=10, dt=0.0125, basis_bandwidth_factor=5, weights_scale=0.1, goal_scale=0.1, relative_goal=True)
t0 = torch.zeros(N, device="cuda"); times = ops.times(t0, mp.dt, T)
w = 0.1 * torch.randn(N, 24, device="cuda", generator=g); y0 = torch.rand(N, 4, device="cuda", generator=g); v0 = torch.zeros(N, 4, device="cuda")
for _ in range(3):
    big.fill_(1.0)
    ops.prodmp_traj(mp, times, w, t0, y0, v0)
# round 2: wide critic (C3 rows, fp32), env rollout kernel, dof-7 trajectories
wide = MLP("ValueFunction", 22, 1, [256, 256], "orthogonal", 1.0, "leaky_relu", None, torch.float32, torch.device("cuda"))
xw = torch.randn(8192, 101, 36, device="cuda", generator=g)[:, :-1, :22]; rw = torch.randn(8192, 100, device="cuda", generator=g)
runw = critic_ops.make_runner(wide)
for _ in range(3):
    big.fill_(1.0)
    runw.epoch(xw, rw, rw, 0.0)
# round 3: the same critic in float64
wide64 = MLP("ValueFunction", 22, 1, [256, 256], "orthogonal", 1.0, "leaky_relu", None, torch.float64, torch.device("cuda"))
runw64 = critic_ops.make_runner(wide64)
xw64, rw64 = xw.double(), rw.double()
for _ in range(3):
    big.fill_(1.0)
    runw64.epoch(xw64, rw64, rw64, 0.0)
del xw64, rw64
# round 4: the float64 128 x 2 policy mean net of box pushing on csrc/pmlp.hip (8192 rows)
from tce_rl_amd import pmlp_ops
pnet = MLP("policy", 22, 63, [128, 128], "orthogonal", 0.01, "leaky_relu", None, torch.float64, torch.device("cuda"))
xp = torch.randn(8192, 22, device="cuda", generator=g, dtype=torch.float64); gp = torch.randn(8192, 63, device="cuda", generator=g, dtype=torch.float64)
for _ in range(3):
    big.fill_(1.0)
    keep = {}
    pmlp_ops.forward(pnet, xp, keep=keep)
    pmlp_ops.backward(pnet, keep, gp)
acts = torch.randn(N, T, 8, device="cuda", generator=g); obs0 = torch.randn(N, 48, device="cuda", generator=g); shift = torch.zeros(48, device="cuda")
for _ in range(3):
    big.fill_(1.0)
    ops.env_rollout(acts, obs0, "reach", 4, 39, 0.0125, 400.0, 40.0, want_states=True, shift=shift, want_moments=True)
mp7 = ProDMP(dtype=torch.float32, device="cuda", num_dof=7, num_basis=8, tau=2.0, alpha_phase=3, alpha=10, dt=0.02, basis_bandwidth_factor=3, weights_scale=0.3, goal_scale=0.3)
n7 = 65536; t07 = torch.zeros(n7, device="cuda"); times7 = ops.times(t07, mp7.dt, 100)
w7 = 0.1 * torch.randn(n7, 63, device="cuda", generator=g); y07 = torch.rand(n7, 7, device="cuda", generator=g); v07 = torch.zeros(n7, 7, device="cuda")
for _ in range(3):
    big.fill_(1.0)
    ops.prodmp_traj(mp7, times7, w7, t07, y07, v07)
torch.cuda.synchronize()
