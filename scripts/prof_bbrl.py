"""One BASELINE configs[3] shard (BBRL, 4096 envs, 100 + 100 epochs) for
rocprofv3 --kernel-trace --stats:  python scripts/prof_bbrl.py [N] [epochs] [iters]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd.config import bbrl_config
from tce_rl_amd.mp_exp import MPExperiment

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
E = int(sys.argv[2]) if len(sys.argv) > 2 else 100
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 4
cfg = bbrl_config(num_env=N, epochs=E)
if os.environ.get("TCE_BB_SMALL") == "0":
    cfg["params"]["agent"]["args"]["small_net_kernels"] = False
exp = MPExperiment()
exp.initialize(cfg, 0, None)
for i in range(iters):
    torch.cuda.synchronize()
    t = time.perf_counter()
    res = exp.agent.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("%d: %.2f ms  sampling %.2f update %.2f" % (
        i, dt * 1e3, res["sampling_time"] * 1e3, res["update_time"] * 1e3),
        flush=True)
