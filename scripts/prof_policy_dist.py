"""scripts/prof_policy.py through the sharded code path: a ONE-rank RCCL world
(TCE_FORCE_DIST=1), policy update alone and the whole overlapped step."""
import sys, os, time, torch
os.environ["TCE_FORCE_DIST"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
os.environ["RANK"], os.environ["WORLD_SIZE"] = "0", "1"
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from tce_rl_amd.config import tce_config
from tce_rl_amd.mp_exp import MPExperiment
overlap = "overlap" in sys.argv
cfg = tce_config("metaworld", num_env=4096, num_basis=5, epochs=50, evaluation_interval=0)
cfg["params"]["agent"]["args"]["overlap_updates"] = overlap
exp = MPExperiment(); exp.initialize(cfg, 0, None)
ag = exp.agent
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = ag.step()
    torch.cuda.synchronize()
    print("step %.1f ms  critic %.1f  policy %.1f" % (1e3 * (time.perf_counter() - t0), 1e3 * r["update_critic_time"], 1e3 * r["update_policy_time"]), flush=True)
ds, _ = ag.sampler.run(training=True, policy=ag.policy, critic=ag.critic)
ds = ag.process_dataset(ds)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ag.update_policy(ds)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"update_policy: host-return {1e3*(t1-t0):.1f} ms, synced {1e3*(t2-t0):.1f} ms", flush=True)
dist.destroy_process_group()
