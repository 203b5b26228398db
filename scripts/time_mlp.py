import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd import critic_ops, mlp_ops
from tce_rl_amd.nn import MLP
torch.manual_seed(0)
mlp = MLP("ValueFunction", 40, 1, [128, 128], "orthogonal", 1.0, "relu", None, torch.float32, torch.device("cuda"))
full = torch.randn(4096, 501, 48, device="cuda"); x = full[:, :-1, :40]; ret = torch.randn(4096, 500, device="cuda")
run = critic_ops.EpochRunner(mlp)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
ms = t(lambda: run.epoch(x, ret, ret, 0.0))
R = 4096 * 500; fl = R * (2 * (40 * 128 + 128 * 128 + 128) * 3)
print(f"fused epoch fwd+bwd: {ms:.3f} ms  -> {fl / ms / 1e9:.1f} TFLOP/s (fp32 MFMA peak 157)")
ms = t(lambda: critic_ops.forward(mlp, x))
print(f"fused forward: {ms:.3f} ms -> {fl / 3 / ms / 1e9:.1f} TFLOP/s")
def lib():
    for p in mlp.parameters(): p.grad = None
    (ret - mlp_ops.forward(mlp, x).squeeze(-1)).pow(2).mean().backward()
ms = t(lib, 5)
print(f"library path epoch: {ms:.3f} ms")
