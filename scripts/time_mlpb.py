"""Times one critic epoch at BASELINE C2 rows (4096 envs x 500 steps, D_in 40)
on the three arithmetic variants of the 128 x 2 critic kernel: exact fp32
(csrc/mlp.hip), split f16 x 2 (csrc/mlp16.hip), three-part bf16 (csrc/mlpb.hip).
usage: python scripts/time_mlpb.py [reps] [max_workgroups]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from tce_rl_amd import critic_ops  # noqa: E402
from test_mlp_gpu import make  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 0
mlp = make(40, "relu", 5)
g = torch.Generator(device="cuda").manual_seed(2)
full = torch.randn(4096, 501, 48, device="cuda", generator=g)
x = full[:, :-1, :40]
ret = torch.randn(4096, 500, device="cuda", generator=g)
for arith in ("f32", "f16x2", "bf16x3"):
    run = critic_ops.EpochRunner(mlp, arith=arith)
    for _ in range(3):
        run.epoch(x, ret, ret, 0.0, max_workgroups=cap)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run.epoch(x, ret, ret, 0.0, max_workgroups=cap)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-7s %.3f ms per epoch  (%.1f TFLOP/s algorithmic)" % (arith, ms, 265.8e9 / ms / 1e9))
