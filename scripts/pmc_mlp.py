"""Launch the fused critic epoch kernel a few times (for rocprofv3 --pmc passes)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd import critic_ops
from tce_rl_amd.nn import MLP
N, T = 4096, 500
g = torch.Generator(device="cuda").manual_seed(0)
torch.manual_seed(0)
mlp = MLP("ValueFunction", 40, 1, [128, 128], "orthogonal", 1.0, "relu", None, torch.float32, torch.device("cuda"))
full = torch.randn(N, T + 1, 48, device="cuda", generator=g); x = full[:, :-1, :40]
r = torch.randn(N, T, device="cuda", generator=g)
run = critic_ops.EpochRunner(mlp, arith=(sys.argv[1] if len(sys.argv) > 1 else "f32"))
for _ in range(3):
    run.epoch(x, r, r, 0.0)
torch.cuda.synchronize()
