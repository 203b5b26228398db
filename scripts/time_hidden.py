"""Policy mean net's hidden layers (39 -> 128 -> 128, 4096 rows) on the fused MLP
kernels, alone on the GPU: forward and backward, with and without a CU budget."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd import critic_ops, _lib
from tce_rl_amd.nn import MLP
mlp = MLP("MeanNet", 39, 24, [128, 128], "orthogonal", 1.0, "relu", None, torch.float32, torch.device("cuda"))
x = torch.randn(4096, 39, device="cuda")
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for budget in (0, 32):
    _lib.call("tce_set_cu_budget", budget)
    h = critic_ops.hidden_forward(mlp, x)
    print("budget", budget, "hidden forward %.1f us" % timeit(lambda: critic_ops.hidden_forward(mlp, x)))
    xr = x.clone().requires_grad_(False)
    def fb():
        hh = critic_ops.hidden_forward(mlp, x)
        hh.sum().backward()
    with torch.enable_grad():
        print("budget", budget, "forward + backward (autograd node) %.1f us" % timeit(fb, 30))
_lib.call("tce_set_cu_budget", 0)
