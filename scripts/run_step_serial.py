"""One config, critic and policy updates one after the other (no stream
overlap): device time of each update alone.
    python scripts/run_step_serial.py <env> <N> <epochs> <iters> <dtype>"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd.config import tce_config
from tce_rl_amd.mp_exp import MPExperiment
env, N, epochs, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dtype = sys.argv[5] if len(sys.argv) > 5 else "float32"
cfg = tce_config(env, num_env=N, epochs=epochs, dtype=dtype)
cfg["params"]["agent"]["args"]["overlap_updates"] = False
exp = MPExperiment(); exp.initialize(cfg, 0, None)
for i in range(iters):
    torch.cuda.synchronize(); t = time.perf_counter()
    res = exp.iterate(cfg, 0, i)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(i, f"{dt*1e3:.1f} ms", {k: round(res[k], 5) for k in ("sampling_time", "update_critic_time", "update_policy_time", "policy_epochs_device_time")}, flush=True)
