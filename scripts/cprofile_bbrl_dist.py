"""cProfile of the sharded (one-rank RCCL world) black-box step of C4: where the host time goes."""
import cProfile, os, pstats, sys, time
os.environ["TCE_FORCE_DIST"] = "1"
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29743")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
import bench
agent = bench.build_config_agent(dict(bench.OTHER_CONFIGS)["C4_bbrl_shard"])
agent.balance_check = None
for _ in range(6):
    agent.step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    agent.step()
torch.cuda.synchronize()
print("ms per step %.2f" % ((time.perf_counter() - t) * 100), flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    agent.step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
dist.destroy_process_group()
