"""cProfile of the sharded (one-rank RCCL world) TCE step at C2: where the host time goes."""
import cProfile, os, pstats, sys, time
os.environ["TCE_FORCE_DIST"] = "1"
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29747")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import bench
agent, cfg = bench.build_agent(4096, seed=0)
for _ in range(8):
    agent.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
t = time.perf_counter()
for _ in range(5):
    agent.step()
torch.cuda.synchronize()
print("ms per step %.2f" % ((time.perf_counter() - t) * 200), flush=True)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
dist.destroy_process_group()
