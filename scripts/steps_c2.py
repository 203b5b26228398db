"""A few C2 steps (BASELINE configs[1]) for a kernel trace:  python scripts/steps_c2.py [iters]"""
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tce_rl_amd.config import tce_config
from tce_rl_amd.mp_exp import MPExperiment
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = tce_config("metaworld", num_env=4096, epochs=50, dtype="float32")
exp = MPExperiment(); exp.initialize(cfg, 0, None)
results = [exp.agent.step() for i in range(iters)]     # metrics read after the loop
torch.cuda.synchronize()
print("done", results[-1]["num_global_steps"])
