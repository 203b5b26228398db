"""ONE torch-CPU oracle step at the HEADLINE size -- BASELINE.json configs[1]:
4096 envs, T 500, ProDMP 5 basis, 50 critic + 50 policy epochs, fp32 -- on every
host core this process may use (BASELINE.md 3 / SURVEY 8d describe the CPU
baseline so; bench.py's in-run `cpu_baseline` is a 64-env sample because this
one takes minutes).  Offline; the result is committed as
profiles/r05_cpu_4096.json and cited by bench.py's cpu_baseline.sample.

    python scripts/cpu_full_size.py [num_env] [out.json] [threads ...]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from oracle.agent_oracle import OracleTCE  # noqa: E402  (checker / baseline only)
from tce_rl_amd.config import tce_config  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
out = sys.argv[2] if len(sys.argv) > 2 else None
try:
    avail = len(os.sched_getaffinity(0))
except AttributeError:
    avail = os.cpu_count() or 1
res = []
# (thread counts: the GPU boxes show 256 cores but give a process a 16-core share;
# 256 threads on that share thrash -- the count that is best of the ones tried
# is the one reported, all are listed)
counts = [int(a) for a in sys.argv[3:]] or sorted({min(avail, 16), min(avail, 64)})
for threads in counts:
    torch.set_num_threads(threads)
    cfg = tce_config("metaworld", num_env=n, num_basis=bench.NUM_BASIS,
                     epochs=bench.EPOCHS, device="cpu")
    o = OracleTCE(cfg["params"], n)
    t = time.perf_counter()
    steps = o.step()                # (no warm-up: one step is minutes)
    dt = time.perf_counter() - t
    res.append({"threads": threads, "seconds_per_step": round(dt, 2),
                "env_steps_per_sec": round(steps / dt, 1)})
    print(res[-1], flush=True)
best = max(res, key=lambda r: r["env_steps_per_sec"])
doc = {"what": "torch-CPU oracle agent.step() (kind 'port') at the headline "
               "workload: %d envs, T 500, nb 5, 50 + 50 epochs, fp32; ONE step "
               "per thread count, no warm-up" % n,
       "cpu": bench.cpu_model(), "visible_cores": avail, "runs": res,
       "value": best["env_steps_per_sec"], "unit": "env-steps/s",
       "cores": best["threads"]}
print(json.dumps(doc))
if out:
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
