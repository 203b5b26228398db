"""Where the wide float64 critic loses its 4 points inside the step (VERDICT r5
item 7): C3 in the reference's dtype (8192 envs, T 100, 22 -> 256 -> 256 -> 1,
float64), critic time per epoch from HIP events on the critic's stream
 * overlapped with the policy stream at several `critic_workgroups`
   (the CUs lent to the policy while it runs),
 * with the adaptive split off and ALL epochs on the reduced grid,
 * serial (overlap_updates = false: every epoch owns the chip).
Prints one JSON line per arrangement."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

spec = dict(dict(bench.OTHER_CONFIGS)[sys.argv[1] if len(sys.argv) > 1
                                       else "C3_box_push_f64"])
rows = spec["num_env"] * 100
E = spec["epochs"]
din, H = 22, 256
flops = 6.0 * (din * H + H * H + H) * rows
peak = bench.F64_MFMA_PEAK_TF if spec["dtype"] == "float64" \
    else bench.F32_MFMA_PEAK_TF
cases = [("overlap wg224 (default)", dict()),
         ("overlap wg240", dict(critic_workgroups=240)),
         ("overlap wg248", dict(critic_workgroups=248)),
         ("overlap wg192", dict(critic_workgroups=192)),
         ("serial (overlap_updates=false)", dict(overlap_updates=False))]
for name, kw in cases:
    agent = bench.build_config_agent(spec)
    for k, v in kw.items():
        setattr(agent, k, v)
    for _ in range(3):
        agent.step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    res = [agent.step() for _ in range(3)]
    torch.cuda.synchronize()
    el = (time.perf_counter() - t) / 3
    crit = sum(r["update_critic_time"] for r in res) / 3
    pol = sum(r["update_policy_time"] for r in res) / 3
    us = crit / E * 1e6
    print(json.dumps({
        "case": name, "ms_per_step": round(el * 1e3, 2),
        "critic_ms": round(crit * 1e3, 2), "policy_ms": round(pol * 1e3, 2),
        "critic_us_per_epoch": round(us, 1),
        "frac_in_step": round(flops / us / 1e6 / peak, 4),
        "critic_split": getattr(agent, "_critic_split", None)}), flush=True)
    agent.close()
    del agent
    torch.cuda.empty_cache()
