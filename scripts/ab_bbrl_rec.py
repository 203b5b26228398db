"""A / B of the black-box row kernels' record row: 19 columns (7 scalars + the 12
KL means of kl_old_new_proj) against the 7 of round 3, C4 shard, alternating
blocks of steps on one agent.
    python scripts/ab_bbrl_rec.py [rounds] [steps per block]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import bench
from tce_rl_amd import _lib, smlp_ops

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
block = int(sys.argv[2]) if len(sys.argv) > 2 else 12
spec = dict(bench.OTHER_CONFIGS)["C4_bbrl_shard"]
agent = bench.build_config_agent(spec)
agent.balance_check = None
real_call = smlp_ops.call
stride = [19]


def call(name, *args):
    if name == "tce_bb_policy_epochs_f32":
        args = list(args)
        i = args.index(19)                 # rec_stride (the only 19 among the arguments)
        args[i] = stride[0]
    return real_call(name, *args)


smlp_ops.call = call
for _ in range(5):
    agent.step()
torch.cuda.synchronize()
res = {7: [], 19: []}
for r in range(rounds):
    for s in (19, 7):
        stride[0] = s
        agent.step()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(block):
            agent.step()
        torch.cuda.synchronize()
        res[s].append((time.perf_counter() - t) / block * 1e3)
for s in (7, 19):
    print("rec_stride %2d: %s  min %.3f ms" % (s, " ".join("%.3f" % x for x in res[s]), min(res[s])))
