"""Step time of BlackBoxAgent on the reference's box-pushing / table-tennis BBRL
documents (tests/golden/resolved/*_bbrl.json: 128 x 2 + 256 x 2 and 256 x 1 +
256 x 1 nets, full covariance, the documents' own epoch counts) at ENVS envs:
hand-written epochs (objective.BBDirectEpoch, pmlp / matrix-core critic) against
the HIP-graph + library-GEMM path they replace (small_net_kernels=False).

    python scripts/time_bbrl_docs.py [ENVS] [hand|graph|both] [STEPS] [box|table]
"""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from tce_rl_amd.mp_exp import MPExperiment

ENVS = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
MODES = {"hand": (True,), "graph": (False,), "both": (True, False)}[
    sys.argv[2] if len(sys.argv) > 2 else "both"]
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 7
ONLY = sys.argv[4] if len(sys.argv) > 4 else ""


def build(doc, hand):
    d = json.load(open(os.path.join(R, "tests", "golden", "resolved", doc + ".json")))
    p = d["params"]
    for blk in p.values():
        blk["args"]["device"] = "cuda"
    sa = p["sampler"]["args"]
    sa.update(num_env_train=ENVS, num_env_test=8, task_specified_metrics=["success"])
    p["agent"]["args"].update(evaluation_interval=0)
    fam = "TableTennis" if "TableTennis" in sa["env_id"] else "BoxPushing"
    defaults = {"BoxPushing": dict(alpha=10, dt=0.02, tau=2.0),
                "TableTennis": dict(alpha=25, dt=0.008, tau=0.75)}[fam]
    for k, v in dict(defaults, alpha_phase=3, basis_bandwidth_factor=3,
                     dtype=p["agent"]["args"]["dtype"], device="cuda").items():
        p["mp"]["args"].setdefault(k, v)
    if "mp" in sa:
        sa["mp"] = p["mp"]
    if "mp" in p["policy"]["args"]:
        p["policy"]["args"]["mp"] = p["mp"]
    cfg = {"name": d["name"], "seed": 0, "iterations": d["iterations"], "params": p}
    torch.manual_seed(0)
    exp = MPExperiment()
    exp.initialize(cfg, 0, None)
    exp.agent.small_net_kernels = hand
    return exp.agent


for doc in ("box_push_random_init_bbrl", "table_tennis_4d_bbrl"):
    if ONLY and not doc.startswith(ONLY):
        continue
    for hand in MODES:
        agent = build(doc, hand)
        ts = []
        for i in range(STEPS):
            torch.cuda.synchronize(); t = time.perf_counter()
            res = agent.step()
            if i in (0, STEPS - 1):
                res = dict(res)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        E = agent.epochs_policy
        print("%s  %s  K %d  epochs %d + %d  paths %s / %s : steps %s ms  (min past the first: %.2f)"
              % (doc, "hand-written" if hand else "graph+library", agent.policy.dim_out, E,
                 agent.epochs_critic, agent._critic_path(),
                 agent._policy_path({"segment_params_L": agent.policy.policy(
                     torch.zeros(2, agent.policy.mean_net.dim_in, device="cuda",
                                 dtype=agent.dtype))[1],
                     "segment_state": torch.zeros(ENVS, agent.policy.mean_net.dim_in,
                                                  device="cuda", dtype=agent.dtype)}),
                 " ".join("%.1f" % t for t in ts), min(ts[1:])), flush=True)
        del agent
