"""Phase cycle stamps of the black-box agent's row kernel (diagnostic build of
smlp.hip with -DSMLP_STAMP in its own .so; the product library is untouched):
    python scripts/smlp_stamps.py [N] [K]
Workgroup 0, wave 0: {setup + tile loads, forward, head, backward, gradients}."""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CS = os.path.join(ROOT, "tce_rl_amd", "csrc")
so = os.path.join(ROOT, "scripts", "variants", "libsmlp_stamp.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
if "--build" in sys.argv or not os.path.exists(so):
    # the stamped smlp.hip + the product build's objects of everything else
    from tce_rl_amd.build import build_library
    build_library(verbose=False)
    obj = os.path.join(CS, "build")
    others = [os.path.join(obj, f) for f in sorted(os.listdir(obj))
              if f.endswith(".o") and f != "smlp.o"]
    st_o = os.path.join(os.path.dirname(so), "smlp_stamp.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3",
                           "-fPIC", "-std=c++17", "-DSMLP_STAMP"] + (["-DSMLP_STAMP_HEAD"] if "--head" in sys.argv else []) + ["-c",
                           os.path.join(CS, "smlp.hip"), "-o", st_o])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950",
                           "-shared", "-fPIC", "-o", so, st_o] + others)
    if "--build" in sys.argv:
        sys.exit(0)
from tce_rl_amd import _lib
_lib.LIB_PATH = so                       # the same prototypes, the stamped kernels
sys.path.insert(0, os.path.join(ROOT, "tests"))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]) if args else 4096
from tce_rl_amd.config import bbrl_config
from tce_rl_amd.mp_exp import MPExperiment
from tce_rl_amd import smlp_ops
cfg = bbrl_config(num_env=N, epochs=10)
exp = MPExperiment()
exp.initialize(cfg, 0, None)
agent = exp.agent
for _ in range(2):
    agent.step()
torch.cuda.synchronize()
lib = _lib.load()
names = ["setup+loads", "forward", "head", "backward", "gradients"]
for net, dout, tag in ((agent.critic.net, 1, "critic"),
                       (agent.policy.mean_net, agent.policy.dim_out, "policy")):
    ws = net.__dict__["_tce_smlp_ws"][("ws", N)]
    H = net.hidden_layers[0]
    P = lib.tce_smlp_num_params(net.dim_in, H, dout)
    PS = (P + dout * dout + 3) // 4 * 4
    grid = min((N + 63) // 64, 1024)
    o = grid * PS + 16 * grid + 24
    st = ws[o:o + 5].cpu().tolist()
    print(tag, {k: int(v) for k, v in zip(names, st)}, "total", int(sum(st)))
