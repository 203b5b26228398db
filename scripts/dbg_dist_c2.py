"""Debug: C2 through a one-rank RCCL world; prints the adopted critic split and per-step device times."""
import os, sys, time
os.environ["TCE_FORCE_DIST"] = "1"
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29745")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import bench
agent, cfg = bench.build_agent(4096, seed=0)
if os.environ.get("DBG_BAL"):
    agent.balance_check = 2            # every other iteration carries the balance check
res = []
for i in range(12):
    r = agent.step()
    res.append(r)
    print("iter", i + 1, "split", agent._critic_split, agent._critic_split_bal, "local", agent._local_split, flush=True)
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(10):
    res.append(agent.step())
torch.cuda.synchronize()
print("ms per step %.2f" % ((time.perf_counter() - t) * 100), "split", agent._critic_split, flush=True)
for r in res[-3:]:
    print({k: round(float(r[k]) * 1e3, 2) for k in ("sampling_time", "update_time", "update_critic_time", "update_policy_time")})
print("kind", agent.dist.exchange_kind())
dist.destroy_process_group()
