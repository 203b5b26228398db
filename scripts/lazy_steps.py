"""Per-step device times of C2 steps with lazy metrics (and the critic split each used)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd.config import tce_config
from tce_rl_amd.mp_exp import MPExperiment
cfg = tce_config("metaworld", num_env=4096, epochs=50, dtype="float32")
exp = MPExperiment(); exp.initialize(cfg, 0, None)
ag = exp.agent
res, splits, walls = [], [], []
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    res.append(ag.step()); splits.append(ag._critic_split); walls.append(time.perf_counter() - t0)
torch.cuda.synchronize(); tot = time.perf_counter() - t0
for i, r in enumerate(res):
    print(i, "split(after)", splits[i], "host returned at %.1f ms" % (walls[i] * 1e3),
          "critic %.2f ms policy %.2f sampling %.2f update %.2f" % (r["update_critic_time"] * 1e3, r["update_policy_time"] * 1e3, r["sampling_time"] * 1e3, r["update_time"] * 1e3))
print("total %.1f ms for %d steps" % (tot * 1e3, len(res)))
