"""Summarise the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate
runs of scripts/pmc_kernels.py) into profiles/<tag>_pmc.json and compact CSVs.

    python scripts/pmc_summarize.py <fetch_counter_collection.csv> \
        <write_counter_collection.csv> <tag>

Correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE counts the
128-B requests of wide coalesced reads at 64 B -> doubled; WRITE_SIZE as is.
Both counters are reported in KiB."""
import csv, json, os, sys, collections

KERNELS = ("gae_dpp_kernel", "mlp_critic_bwd_kernel", "mlp_critic_bwd16_kernel",
           "mlp_critic_fwd_kernel", "prodmp_traj_rows_kernel<float, 4", "prodmp_traj_rows_kernel<float, 7",
           "mlpw_chain_kernel<float", "mlpw_grad_kernel<float", "mlpw_chain_kernel<double",
           "mlpw_grad_kernel<double", "env_rollout_kernel<float, 12",
           "smlp_epoch_kernel<32, 1, 1", "smlp_epoch_kernel<32, 4, 2", "smlp_reduce_kernel")
# names bench.py looks up (round 2 keys) -> round 3 keys
ALIAS = {"mlpw_chain_kernel": "mlpw_chain_kernel<float", "mlpw_grad_kernel": "mlpw_grad_kernel<float",
         "env_rollout_kernel": "env_rollout_kernel<float, 12"}
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path, counter):
    vals = collections.defaultdict(list)
    rows = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for k in KERNELS:
            if k in r["Kernel_Name"]:
                vals[k].append(float(r["Counter_Value"]))
                rows.append((r["Kernel_Name"][:100], counter, r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in vals.items()}, rows


def main():
    fpath, wpath, tag = sys.argv[1:4]
    fetch, frows = per_kernel(fpath, "FETCH_SIZE")
    write, wrows = per_kernel(wpath, "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) "
                     "-- python3 scripts/pmc_kernels.py; C2 shapes; caches flushed with a "
                     "1 GiB fill before every launch; scripts/pmc_summarize.py",
           "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests of wide coalesced "
                         "reads at 64 B: MI355X_MICROARCH.md, HBM section); WRITE_SIZE as "
                         "is; both in KiB",
           "kernels": {}}
    for k in KERNELS:
        if k in fetch and k in write:
            fb, wb = int(fetch[k] * 1024 * 2), int(write[k] * 1024)
            out["kernels"][k] = {"fetch_bytes_corrected": fb, "write_bytes": wb,
                                 "traffic_bytes": fb + wb}
    for a, k in ALIAS.items():
        if k in out["kernels"]:
            out["kernels"][a] = out["kernels"][k]
    with open(os.path.join(REPO, "profiles", tag + "_pmc.json"), "w") as f:
        json.dump(out, f, indent=1)
    for name, rows in (("FETCH_SIZE", frows), ("WRITE_SIZE", wrows)):
        with open(os.path.join(REPO, "profiles", "%s_pmc_%s.csv" % (tag, name)), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value_KiB"])
            w.writerows(rows)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
