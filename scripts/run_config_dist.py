"""scripts/run_config.py inside a ONE-rank RCCL world with TCE_FORCE_DIST=1: the
sharded code path (collectives, un-fused optimizer steps) of a `configs` entry on
one GPU.    python scripts/run_config_dist.py C4_bbrl_shard [steps] [warmup]"""
import json, os, sys
os.environ["TCE_FORCE_DIST"] = "1"
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29731")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
import bench  # noqa: E402
name = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
warmup = int(sys.argv[3]) if len(sys.argv) > 3 else 2
spec = dict(bench.OTHER_CONFIGS)[name]
out = bench.run_config(name, spec, steps, warmup)
from tce_rl_amd import dist as tdist
print(json.dumps({k: v for k, v in out.items() if k in ("ms_per_step", "balance_check_iteration_ms", "policy_updates_per_sec")}), tdist.stats())
dist.destroy_process_group()
