"""The policy net's hidden forward / the pair kernels' neighbours while the critic's
persistent grid (224 workgroups) runs on another stream: per-launch device time."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd import critic_ops, _lib
from tce_rl_amd.nn import MLP
dev = torch.device("cuda")
crit = MLP("ValueFunction", 39, 1, [128, 128], "orthogonal", 1.0, "relu", None, torch.float32, dev)
pol = MLP("MeanNet", 39, 24, [128, 128], "orthogonal", 1.0, "relu", None, torch.float32, dev)
xs = torch.randn(4096, 501, 47, device=dev)[:, :-1, :39]
ret = torch.randn(4096, 500, device=dev)
x = torch.randn(4096, 39, device=dev)
run = critic_ops.make_runner(crit)
side = torch.cuda.Stream()
wg = int(sys.argv[1]) if len(sys.argv) > 1 else 224
for budget in (0, 32):
    _lib.call("tce_set_cu_budget", budget)
    for _ in range(3):
        run.epoch(xs, ret, ret, 0.0, wg); critic_ops.hidden_forward(pol, x)
    torch.cuda.synchronize()
    for _ in range(12):
        run.epoch(xs, ret, ret, 0.0, wg)
    evs = []
    with torch.cuda.stream(side):
        torch.cuda._sleep(3_000_000)                     # let the critic get going
        for _ in range(40):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); critic_ops.hidden_forward(pol, x); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    print("critic grid %d, budget %d: hidden forward beside it: median %.1f us, min %.1f, max %.1f" % (wg, budget, ts[len(ts) // 2], ts[0], ts[-1]))
_lib.call("tce_set_cu_budget", 0)
