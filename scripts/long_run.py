"""Long C2 run under host run-ahead: metrics read every 50 iterations only; prints
reward / critic loss / memory so that drift, NaNs or a leak would show."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd.config import tce_config
from tce_rl_amd.mp_exp import MPExperiment
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 600
cfg = tce_config("metaworld", num_env=4096, num_basis=5, epochs=50, evaluation_interval=0, iterations=iters)
exp = MPExperiment(); exp.initialize(cfg, 0, None)
ag = exp.agent
t = time.perf_counter()
for i in range(iters):
    res = ag.step()
    if i % 50 == 49 or i == iters - 1:
        print(i, "reward %.2f" % res["exploration_episode_reward_mean"], "critic_loss %.3f" % res["critic_loss_mean"],
              "kl %.2e" % res["projection_proj_old_cov_diff_mean"], "split", ag._critic_split,
              "mem %.2f GB (max %.2f)" % (torch.cuda.memory_allocated() / 1e9, torch.cuda.max_memory_allocated() / 1e9),
              "%.1f ms/it" % ((time.perf_counter() - t) / (i + 1) * 1e3), flush=True)
