"""Step time of a config against the persistent critic grid size used while the
policy stream runs beside it.  python scripts/tune_wg.py <env> <N> <dtype> wg1 wg2 ..."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd.config import tce_config
from tce_rl_amd.mp_exp import MPExperiment
env, N, dtype = sys.argv[1], int(sys.argv[2]), sys.argv[3]
for wg in map(int, sys.argv[4:]):
    cfg = tce_config(env, num_env=N, epochs=50, dtype=dtype, num_basis=int(os.environ.get("NB", "8")))
    cfg["params"]["agent"]["args"]["critic_workgroups"] = wg
    exp = MPExperiment(); exp.initialize(cfg, 0, None)
    ts = []
    for i in range(5):
        torch.cuda.synchronize(); t = time.perf_counter()
        res = exp.iterate(cfg, 0, i)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print("wg %3d: %.1f ms (critic %.1f, policy %.1f, split %d)" % (wg, min(ts[2:]) * 1e3, res["update_critic_time"] * 1e3,
          res["update_policy_time"] * 1e3, exp.agent._critic_split), flush=True)
    del exp
