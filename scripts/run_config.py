"""A few steps of one entry of bench.py's `configs` block (for rocprofv3):
    python scripts/run_config.py C3_box_push_f32 [steps] [warmup]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

name = sys.argv[1]
nums = [a for a in sys.argv[2:] if a.isdigit()]
steps = int(nums[0]) if len(nums) > 0 else 3
warmup = int(nums[1]) if len(nums) > 1 else 2
spec = dict(dict(bench.OTHER_CONFIGS)[name])
if "nofloor" in sys.argv[2:]:          # (traces: no second, 64-env agent in the run)
    spec["chain_floor"] = False
out = bench.run_config(name, spec, steps, warmup)
print(json.dumps({k: v for k, v in out.items() if k != "workload"}))
