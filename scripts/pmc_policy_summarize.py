"""Per-launch averages of the SQ counters of scripts/pmc_policy.sh for the
policy-epoch kernels -> profiles/<tag>_pmc_policy.json
    python scripts/pmc_policy_summarize.py gpurun_out/pmc_pol <tag>"""
import csv, glob, json, os, sys, collections
root, tag = sys.argv[1], sys.argv[2]
KERNELS = ("pair_env_kernel<float, true>", "pair_env_kernel<float, false>", "kl_cov_proj_fwd_kernel", "kl_cov_proj_bwd_kernel",
           "kl_shared_env_kernel", "kl_shared_mat_kernel", "vec_env_shared_kernel<float, 1, true>", "pair_prep_kernel")
acc = {k: collections.defaultdict(list) for k in KERNELS}
for path in sorted(glob.glob(os.path.join(root, "p*", "p_counter_collection.csv"))):
    for r in csv.DictReader(open(path)):
        for k in KERNELS:
            if k in r["Kernel_Name"]:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"source": "rocprofv3 --pmc (4 passes of 4 SQ / GRBM counters, scripts/pmc_policy.sh) -- python3 scripts/prof_policy.py: "
                 "C2 policy epochs alone (4096 envs, K 24); averages per launch, summed over all SEs / XCDs",
       "kernels": {}}
for k, d in acc.items():
    if not d:
        continue
    e = {c: sum(v) / len(v) for c, v in d.items()}
    gui = e.get("GRBM_GUI_ACTIVE", 0.0) / 8          # per XCD
    if gui and "SQ_WAVES" in e:
        e["derived"] = {
            "gpu_cycles_per_launch": gui,
            "waves": e["SQ_WAVES"],
            "valu_insts_per_wave": e.get("SQ_INSTS_VALU", 0) / e["SQ_WAVES"],
            "lds_insts_per_wave": e.get("SQ_INSTS_LDS", 0) / e["SQ_WAVES"],
            "vmem_rd_insts_per_wave": e.get("SQ_INSTS_VMEM_RD", 0) / e["SQ_WAVES"],
            "wave_cycles_per_wave": e.get("SQ_WAVE_CYCLES", 0) / e["SQ_WAVES"],
            # 1024 SIMDs on the chip; SQ_ACTIVE_INST_VALU counts cycles a SIMD issues VALU work
            "valu_busy_fraction_of_chip": e.get("SQ_ACTIVE_INST_VALU", 0) / (gui * 1024) if gui else None,
            "wave_waiting_fraction": e.get("SQ_WAIT_ANY", 0) / e["SQ_WAVE_CYCLES"] if e.get("SQ_WAVE_CYCLES") else None,
            "wave_waiting_on_lds_fraction": e.get("SQ_WAIT_INST_LDS", 0) / e["SQ_WAVE_CYCLES"] if e.get("SQ_WAVE_CYCLES") else None,
        }
    out["kernels"][k] = e
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
json.dump(out, open(os.path.join(repo, "profiles", tag + "_pmc_policy.json"), "w"), indent=1)
for k, e in out["kernels"].items():
    print(k, json.dumps(e.get("derived", {}), indent=None))
