"""C2 steps with a given critic grid beside the policy stream (kernel trace):
python scripts/steps_c2_wg.py WG [iters]"""
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tce_rl_amd.config import tce_config
from tce_rl_amd.mp_exp import MPExperiment
wg = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = tce_config("metaworld", num_env=4096, epochs=50, dtype="float32")
cfg["params"]["agent"]["args"]["critic_workgroups"] = wg
exp = MPExperiment(); exp.initialize(cfg, 0, None)
for i in range(iters):
    res = exp.iterate(cfg, 0, i)
torch.cuda.synchronize()
print("done", res["update_policy_time"])
