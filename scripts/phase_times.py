"""Per-phase device times of C2 steps, plain or through a one-rank RCCL world (TCE_FORCE_DIST=1):
    [TCE_FORCE_DIST=1] python scripts/phase_times.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
forced = os.environ.get("TCE_FORCE_DIST") == "1"
if forced:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29749")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import bench
agent, cfg = bench.build_agent(4096, seed=0)
for _ in range(8):
    agent.step()
torch.cuda.synchronize()
res = []
t = time.perf_counter()
for _ in range(20):
    res.append(agent.step())
torch.cuda.synchronize()
wall = (time.perf_counter() - t) / 20 * 1e3
keys = ("sampling_time", "process_dataset_time", "update_time", "update_critic_time", "update_policy_time")
avg = {k: sum(float(r[k]) for r in res) / len(res) * 1e3 for k in keys}
print("dist" if forced else "plain", "wall %.2f ms" % wall, {k: round(v, 2) for k, v in avg.items()},
      "split", agent._critic_split, flush=True)
# one balance-check iteration, timed by itself
agent.balance_check = 2
while agent.num_iterations % 2 != 0:
    agent.step()
torch.cuda.synchronize()
t = time.perf_counter()
r = agent.step()
torch.cuda.synchronize()
print("balance iteration wall %.2f ms" % ((time.perf_counter() - t) * 1e3),
      {k: round(float(r[k]) * 1e3, 2) for k in keys}, "bal split", agent._critic_split_bal, flush=True)
if forced:
    dist.destroy_process_group()
