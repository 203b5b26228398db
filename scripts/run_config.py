"""A few steps of one entry of bench.py's `configs` block (for rocprofv3):
    python scripts/run_config.py C3_box_push_f32 [steps] [warmup]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

name = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
warmup = int(sys.argv[3]) if len(sys.argv) > 3 else 2
spec = dict(bench.OTHER_CONFIGS)[name]
out = bench.run_config(name, spec, steps, warmup)
print(json.dumps({k: v for k, v in out.items() if k != "workload"}))
