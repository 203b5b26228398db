import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd import ops
from tce_rl_amd.mp import ProDMP
for name, cfg, N, T in (("C2 nb5", dict(num_dof=4, num_basis=5, tau=5, alpha_phase=3, alpha=10, dt=0.0125, basis_bandwidth_factor=5, weights_scale=0.1, goal_scale=0.1, relative_goal=True), 4096, 500),
                        ("C3 dof7", dict(num_dof=7, num_basis=8, tau=2.0, alpha_phase=3, alpha=10, dt=0.02, basis_bandwidth_factor=3, weights_scale=0.3, goal_scale=0.3), 8192, 100)):
    mp = ProDMP(dtype=torch.float32, device="cuda", **cfg)
    K = mp.num_dof * mp.num_basis_g
    t0 = torch.zeros(N, device="cuda"); times = ops.times(t0, mp.dt, T)
    w = 0.1 * torch.randn(N, K, device="cuda"); y0 = torch.rand(N, mp.num_dof, device="cuda"); v0 = torch.zeros(N, mp.num_dof, device="cuda")
    for _ in range(20): ops.prodmp_traj(mp, times, w, t0, y0, v0)
    torch.cuda.synchronize()
    print(name, "done")
