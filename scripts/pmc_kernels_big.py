"""The HBM kernels at 8 x the C2 env count (past the 256 MB Infinity Cache) and
the dof-7 rows, one size per kernel name, for rocprofv3 --kernel-trace /
--pmc passes (per-size rows for the 8 x cases)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd import ops
from tce_rl_amd.mp import ProDMP

N, T = 8 * 4096, 500
g = torch.Generator(device="cuda").manual_seed(0)
r = torch.randn(N, T, device="cuda", generator=g)
v = torch.randn(N, T + 1, device="cuda", generator=g)
d = torch.zeros(N, T, dtype=torch.bool, device="cuda")
d[:, -1] = True
tl = torch.zeros_like(d)
big = torch.empty(1 << 28, device="cuda")
for _ in range(5):
    big.fill_(1.0)
    ops.gae(r, v, d, tl, 1.0, 0.95, True)
mp = ProDMP(dtype=torch.float32, device="cuda", num_dof=4, num_basis=5, tau=5,
            alpha_phase=3, alpha=10, dt=0.0125, basis_bandwidth_factor=5,
            weights_scale=0.1, goal_scale=0.1, relative_goal=True)
t0 = torch.zeros(N, device="cuda")
times = ops.times(t0, mp.dt, T)
w = 0.1 * torch.randn(N, 24, device="cuda", generator=g)
y0 = torch.rand(N, 4, device="cuda", generator=g)
v0 = torch.zeros(N, 4, device="cuda")
for _ in range(5):
    big.fill_(1.0)
    ops.prodmp_traj(mp, times, w, t0, y0, v0)
torch.cuda.synchronize()
