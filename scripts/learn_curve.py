"""Short training run on the synthetic reach-like env: prints the exploration /
evaluation episode reward every few iterations (learning sanity check)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd.config import tce_config
from tce_rl_amd.mp_exp import MPExperiment
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 20
cfg = tce_config("metaworld", num_env=N, num_basis=5, epochs=epochs, evaluation_interval=0, iterations=iters)
if len(sys.argv) > 4:                      # critic arithmetic: f32 | bf16x3 | f16x2
    cfg["params"]["agent"]["args"]["critic_arith"] = sys.argv[4]
exp = MPExperiment(); exp.initialize(cfg, 0, None)
t = time.perf_counter()
for i in range(iters):
    res = exp.iterate(cfg, 0, i)
    if i % 5 == 0 or i == iters - 1:
        print(i, "reward %.2f" % res["exploration_episode_reward_mean"], "critic_loss %.2f" % res["critic_loss_mean"],
              "entropy %.2f" % res["entropy_mean"], "kl %.2e" % res["projection_proj_old_cov_diff_mean"], flush=True)
print("%.1f s" % (time.perf_counter() - t))
