"""GAE scan: warm (back to back) and cold (after a 1 GiB fill) device time at
4096 and 32768 envs, per prefetch depth (TCE_GAE_PF = 4 | 8 read at the first
launch: one process per depth)."""
import os, sys, subprocess, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) == 1:
    for pf in ("0", "4", "8"):
        subprocess.run([sys.executable, __file__, pf], env=dict(os.environ, TCE_GAE_PF=pf))
    sys.exit(0)
import bench
from tce_rl_amd import ops
for n, T in ((4096, 500), (32768, 500), (32768, 350), (8192, 100)):
    r = torch.randn(n, T, device="cuda"); v = torch.randn(n, T + 1, device="cuda")
    d = torch.zeros(n, T, dtype=torch.bool, device="cuda"); d[:, -1] = True
    tl = torch.zeros_like(d)
    f = lambda: ops.gae(r, v, d, tl, 1.0, 0.95, True)
    alg = n * T * 18 + n * 4
    us, usc = bench.kernel_time_us(f), bench.kernel_time_cold_us(f, launches=9)
    print("PF %s  N %6d T %d: warm %7.1f us (%.3f of 8 TB/s)  cold %7.1f us (%.3f)"
          % (sys.argv[1], n, T, us, alg / us / 8e6, usc, alg / usc / 8e6), flush=True)
