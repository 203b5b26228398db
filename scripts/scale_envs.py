"""env-steps/s of the C2 workload as a function of the envs per GPU."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd.config import tce_config
from tce_rl_amd.mp_exp import MPExperiment
for N in [int(a) for a in sys.argv[1:]] or (1024, 4096, 16384):
    cfg = tce_config("metaworld", num_env=N, epochs=50, num_basis=5, evaluation_interval=0)
    exp = MPExperiment(); exp.initialize(cfg, 0, None)
    for i in range(5):
        torch.cuda.synchronize(); t = time.perf_counter()
        res = exp.iterate(cfg, 0, i)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"N={N}: {dt*1e3:.1f} ms/step  {N*500/dt/1e6:.2f} M env-steps/s  critic {res['update_critic_time']*1e3:.0f} ms policy {res['update_policy_time']*1e3:.0f} ms  mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB", flush=True)
    del exp; torch.cuda.empty_cache()
