"""Experiment: the C2 critic (D_in 39 -> 128 -> 128 -> 1, relu, 4096 x 500 rows) on
the two-launch wide kernels (variant build with -DMLPW_TRY_H128) against the
fused two-role kernel of csrc/mlp.hip.
    MLPW_EXTRA=-DMLPW_TRY_H128 python scripts/mlpw_variant.py 8 32 6 1 --build-only --rebuild
    TCE_HIP_LIB=scripts/variants/libmlpw_w8_pu32_k6_g1_mlpw_try_h128.so python scripts/time_mlpw128.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd import critic_ops
from tce_rl_amd.nn import MLP
N, T, din, H = 4096, 500, 39, 128
torch.manual_seed(0)
mlp = MLP("ValueFunction", din, 1, [H, H], "orthogonal", 1.0, "leaky_relu", None, torch.float32, torch.device("cuda"))
x = torch.randn(N, T + 1, din + 8, device="cuda")[:, :-1, :din]
ret = torch.randn(N, T, device="cuda")
fl = N * T * 6.0 * (din * H + H * H + H)
narrow = critic_ops.EpochRunner(mlp)
critic_ops.narrow_supported = lambda m: False          # let the wide runner take the 128-wide net
for name, run in (("two-role (mlp.hip)", narrow), ("two-launch (mlpw)", critic_ops.WideEpochRunner(mlp))):
    for _ in range(3):
        run.epoch(x, ret, ret, 0.0)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(4):
        s.record()
        for _ in range(10):
            run.epoch(x, ret, ret, 0.0)
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 10)
    print("%-22s %.3f ms / epoch -> %.1f TFLOP/s = %.1f %% of 157.3" % (name, best, fl / best / 1e9, 100 * fl / best / 1e9 / 157.3))
