import sys, torch
sys.path.insert(0, "/root/repo")
from tce_rl_amd import critic_ops
from tce_rl_amd.nn import MLP
from tce_rl_amd.optim import FlatAdam
torch.manual_seed(0)
mlp = MLP("ValueFunction", 40, 1, [128, 128], "orthogonal", 1.0, "relu", None, torch.float32, torch.device("cuda"))
ref = [p.detach().double().cpu().clone().requires_grad_(True) for p in mlp.parameters()]
ref32 = [p.detach().cpu().clone().requires_grad_(True) for p in mlp.parameters()]
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(256, 500, 48, device="cuda", generator=g)[..., :40]
ret = 10 * torch.randn(256, 500, device="cuda", generator=g)
opt = FlatAdam(list(mlp.parameters()), lr=3e-4)
run = critic_ops.EpochRunner(mlp, opt.flat_grad)
o64 = torch.optim.Adam(ref, lr=3e-4); o32 = torch.optim.Adam(ref32, lr=3e-4)
import torch.nn.functional as F
def cpu_epoch(ws, o, dt):
    h = x.reshape(-1, 40).cpu().to(dt)
    h = F.relu(F.linear(h, ws[0], ws[1])); h = F.relu(F.linear(h, ws[2], ws[3]))
    v = F.linear(h, ws[4], ws[5]).squeeze(-1)
    loss = (ret.reshape(-1).cpu().to(dt) - v).pow(2).mean()
    o.zero_grad(); loss.backward(); o.step(); return loss.item()
for e in range(50):
    s = run.epoch(x, ret, ret, 0.0, adam=opt)
    l64 = cpu_epoch(ref, o64, torch.float64); l32 = cpu_epoch(ref32, o32, torch.float32)
    if e % 10 == 9:
        d = max(((p.detach().cpu().double() - r.detach()).abs().max() / r.detach().abs().max()).item() for p, r in zip(mlp.parameters(), ref))
        d32 = max(((p.detach().double() - r.detach()).abs().max() / r.detach().abs().max()).item() for p, r in zip(ref32, ref))
        print(e + 1, "loss gpu %.5f cpu64 %.5f cpu32 %.5f" % (s[0].item(), l64, l32), "max rel param diff: gpu-vs-fp64 %.2e, torch-fp32-vs-fp64 %.2e" % (d, d32), flush=True)
