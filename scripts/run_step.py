import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd.config import tce_config
from tce_rl_amd.mp_exp import MPExperiment
env = sys.argv[1] if len(sys.argv) > 1 else "metaworld"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 2
dtype = sys.argv[5] if len(sys.argv) > 5 else "float32"
cfg = tce_config(env, num_env=N, epochs=epochs, dtype=dtype)
exp = MPExperiment(); exp.initialize(cfg, 0, None)
for i in range(iters):
    torch.cuda.synchronize(); t = time.perf_counter()
    res = exp.iterate(cfg, 0, i)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    keys = ["sampling_time", "process_dataset_time", "update_critic_time", "update_policy_time", "critic_loss_mean", "surrogate_loss_mean", "trust_region_loss_mean", "entropy_mean", "exploration_episode_reward_mean", "projection_new_old_cov_diff_mean", "projection_proj_old_cov_diff_mean", "projection_proj_old_mean_diff_mean", "policy_grad_norm_mean"]
    print(i, f"{dt*1e3:.1f} ms", {k: (round(res[k], 6) if k in res else None) for k in keys}, flush=True)
print("steps/s", res["num_global_steps"] / (i + 1) / dt)
