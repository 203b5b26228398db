"""Split-f16 critic kernel vs fp64 / fp32 references: error table + timing."""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
from test_mlp_gpu import make, torch_ref
from tce_rl_amd import critic_ops

for act, din, N, T in [("relu", 40, 7, 33), ("relu", 21, 5, 64), ("tanh", 40, 64, 50), ("softplus", 33, 9, 20)]:
    mlp = make(din, act, 0)
    g = torch.Generator(device="cuda").manual_seed(1)
    full = torch.randn(N, T + 1, din + 8, device="cuda", generator=g)
    x = full[:, :-1, :din]
    ret = torch.randn(N, T, device="cuda", generator=g) * 3
    old = torch.randn(N, T, device="cuda", generator=g)
    v64, l64, g64 = torch_ref(mlp, x.reshape(-1, din), ret.reshape(-1), old.reshape(-1), 0.0, torch.float64)
    v32, l32, g32 = torch_ref(mlp, x.reshape(-1, din), ret.reshape(-1), old.reshape(-1), 0.0, torch.float32)
    for arith in ("f32", "f16x2"):
        run = critic_ops.EpochRunner(mlp, arith=arith)
        st = run.epoch(x, ret, old, 0.0).cpu()
        errs = []
        for p, a, b in zip(mlp.parameters(), g64, g32):
            errs.append("%.1e/%.1e" % ((p.grad.double() - a).abs().max().item() / a.abs().max().item(),
                                       (b.double() - a).abs().max().item() / a.abs().max().item()))
        print(act, din, N * T, arith, "loss %.7f ref %.7f" % (st[0].item(), l64.item()), " ".join(errs), flush=True)

# timing at the C2 shape
mlp = make(40, "relu", 5)
g = torch.Generator(device="cuda").manual_seed(2)
full = torch.randn(4096, 501, 48, device="cuda", generator=g)
x = full[:, :-1, :40]
ret = torch.randn(4096, 500, device="cuda", generator=g)
for arith in ("f32", "f16x2"):
    run = critic_ops.EpochRunner(mlp, arith=arith)
    for _ in range(3):
        run.epoch(x, ret, ret, 0.0)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(20):
        run.epoch(x, ret, ret, 0.0)
    torch.cuda.synchronize()
    print(arith, "ms/epoch %.3f" % ((time.time() - t0) / 20 * 1e3), flush=True)
