"""Resolve the reference's experiment files (mprl/config/<task>/<tcp|bbrl>/
entire/local.yaml, which import shared.yaml) the way cw2 resolves them and
write the resolved VALUES -- the ``params`` document plus the top-level
iteration / checkpoint keys -- as JSON fixtures under tests/golden/resolved/.

Runs in the build container only (needs /root/reference; reads the YAML files
as data through tce_rl_amd.mp_exp.load_config):

    python tests/golden/make_resolved_cfg.py

The fixtures drive tests/test_config_cpu.py (tce_rl_amd.config's hand-typed
hyper-parameters == the reference's) and tests/test_agent_gpu.py (one GPU
agent.step() per task family from the reference's resolved document)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
REF = "/root/reference/mprl/config"
OUT = os.path.join(HERE, "resolved")


def main():
    from tce_rl_amd import mp_exp
    os.makedirs(OUT, exist_ok=True)
    n = 0
    for task in sorted(os.listdir(REF)):
        for algo in ("tcp", "bbrl"):
            path = os.path.join(REF, task, algo, "entire", "local.yaml")
            if not os.path.exists(path):
                continue
            cfg = mp_exp.load_config(path)
            doc = {"source": os.path.relpath(path, "/root/reference"),
                   "name": cfg.get("name"),
                   "iterations": cfg.get("iterations"),
                   "num_checkpoints": cfg.get("num_checkpoints"),
                   "seed": cfg.get("seed"),
                   "verbose_level": cfg.get("verbose_level"),
                   "params": cfg["params"]}
            with open(os.path.join(OUT, "%s_%s.json" % (task, algo)), "w") as f:
                json.dump(doc, f, indent=1, sort_keys=True)
            n += 1
    print("wrote %d resolved documents to %s" % (n, OUT))


if __name__ == "__main__":
    main()
