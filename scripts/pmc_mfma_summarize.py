"""Matrix-pipe utilisation of the MFMA-bound kernels from one rocprofv3 --pmc
pass (SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES over
scripts/pmc_kernels.py) -> profiles/<tag>_pmc_mfma.json

    python scripts/pmc_mfma_summarize.py gpurun_out/pmc4/MFMA <tag>

mfma_busy = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs):
the share of the launch during which a SIMD's matrix pipe is executing (both
counters are sums over the chip; MI355X_MICROARCH.md, constants table:
SQ_VALU_MFMA_BUSY_CYCLES counts cycles, GRBM_GUI_ACTIVE is summed over the 8 XCDs)."""
import collections
import csv
import glob
import json
import os
import sys

root, tag = sys.argv[1], sys.argv[2]
KERNELS = ("mlp_critic_bwd_kernel", "mlp_critic_bwd16_kernel", "mlp_critic_fwd_kernel",
           "mlpw_chain_kernel<float", "mlpw_grad_kernel<float", "mlpw_chain_kernel<double",
           "mlpw_grad_kernel<double", "pmlp_fwd_kernel<double", "pmlp_bwd_kernel<double",
           "pmlp_fwd_kernel<float", "pmlp_bwd_kernel<float")
acc = {k: collections.defaultdict(list) for k in KERNELS}
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        for k in KERNELS:
            if k in r["Kernel_Name"]:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES "
                 "-- python3 scripts/pmc_kernels.py (scripts/pmc_passes_r04.sh); each kernel "
                 "alone on the chip after a 1 GiB cache-flushing fill; means over 3 launches",
       "kernels": {}}
for k, d in acc.items():
    if not d:
        continue
    e = {c: sum(v) / len(v) for c, v in d.items()}
    gui = e.get("GRBM_GUI_ACTIVE", 0.0) / 8
    if gui:
        e["mfma_busy"] = round(e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024 / gui, 4)
        e["kernel_cycles_per_xcd"] = gui
    out["kernels"][k] = e
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
json.dump(out, open(os.path.join(repo, "profiles", tag + "_pmc_mfma.json"), "w"), indent=1)
for k, e in out["kernels"].items():
    print(k, e.get("mfma_busy"))
