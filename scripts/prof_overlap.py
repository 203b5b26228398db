"""Two overlapped C2 steps after warm-up (for rocprofv3 --kernel-trace)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd.config import tce_config
from tce_rl_amd.mp_exp import MPExperiment
wg = int(sys.argv[1]) if len(sys.argv) > 1 else 224
cfg = tce_config("metaworld", num_env=4096, num_basis=5, epochs=50, evaluation_interval=0)
cfg["params"]["agent"]["args"]["critic_workgroups"] = wg
if os.environ.get("CRITIC_ARITH"):
    cfg["params"]["agent"]["args"]["critic_arith"] = os.environ["CRITIC_ARITH"]
exp = MPExperiment(); exp.initialize(cfg, 0, None)
for i in range(4):
    torch.cuda.synchronize(); t = time.perf_counter()
    res = exp.iterate(cfg, 0, i)
    torch.cuda.synchronize(); print(f"{(time.perf_counter()-t)*1e3:.1f} ms", flush=True)
