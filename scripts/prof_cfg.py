"""One step of a config (for rocprofv3 --kernel-trace): env N epochs dtype"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd.config import tce_config
from tce_rl_amd.mp_exp import MPExperiment
env, N, ep, dtype = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
cfg = tce_config(env, num_env=N, epochs=ep, dtype=dtype, evaluation_interval=0)
exp = MPExperiment(); exp.initialize(cfg, 0, None)
for i in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    exp.iterate(cfg, 0, i)
    torch.cuda.synchronize(); print(f"{(time.perf_counter()-t)*1e3:.1f} ms", flush=True)
