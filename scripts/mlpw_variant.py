"""Time one wide critic epoch (fp32, C3 shape) with another chain-kernel shape:
    python scripts/mlpw_variant.py WAVES PU KPG WGS [--build-only]
builds mlpw_f32.hip with -DMLPW_F32_WAVES/PU/WGS (+ -DMLPW_F32_KPG) into its own
.so (scripts/variants/) next to the product build's other objects and times it
with HIP events (chain + gradient + finish; results are NOT checked here)."""
import os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CS = os.path.join(ROOT, "tce_rl_amd", "csrc")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
W, PU, KPG, WGS = (int(a) for a in args[:4])
extra = [a for a in os.environ.get("MLPW_EXTRA", "").split() if a]   # e.g. "-DMLPW_STASH_AFTER"
tag = "w%d_pu%d_k%d_g%d" % (W, PU, KPG, WGS) + "".join("_" + e.lstrip("-D").lower().replace("=", "") for e in extra) + \
    ("_f64" if "--f64" in sys.argv else "")
so = os.path.join(ROOT, "scripts", "variants", "libmlpw_%s.so" % tag)
os.makedirs(os.path.dirname(so), exist_ok=True)
if not os.path.exists(so) or "--rebuild" in sys.argv:
    from tce_rl_amd.build import build_library
    build_library(verbose=False)
    obj = os.path.join(CS, "build")
    f64 = "--f64" in sys.argv                  # the flags go to mlpw_f64.hip as well
    mine = ["mlpw_f32.o"] + (["mlpw_f64.o"] if f64 else [])
    others = [os.path.join(obj, f) for f in sorted(os.listdir(obj))
              if f.endswith(".o") and f not in mine]
    objs = []
    for src in (["mlpw_f32", "mlpw_f64"] if f64 else ["mlpw_f32"]):
        o = os.path.join(os.path.dirname(so), "%s_%s.o" % (src, tag))
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC",
                               "-std=c++17", "-DMLPW_F32_WAVES=%d" % W, "-DMLPW_F32_PU=%d" % PU,
                               "-DMLPW_F32_WGS=%d" % WGS, "-DMLPW_F32_KPG=%d" % KPG,
                               ] + ([] if os.environ.get("MLPW_ALL_ACTS") else ["-DMLPW_ONLY_LEAKY"]) +
                              extra + ["-c", os.path.join(CS, src + ".hip"), "-o", o])
        objs.append(o)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC",
                           "-o", so] + objs + others)
if "--build-only" in sys.argv:
    sys.exit(0)
from tce_rl_amd import _lib
_lib.LIB_PATH = so
from tce_rl_amd import critic_ops
from tce_rl_amd.nn import MLP
N, T, din, H = 8192, 100, 22, 256
torch.manual_seed(0)
dt, peak = (torch.float64, 78.6) if "--f64" in sys.argv else (torch.float32, 157.3)
mlp = MLP("ValueFunction", din, 1, [H, H], "orthogonal", 1.0, "leaky_relu", None,
          dt, torch.device("cuda"))
x = torch.randn(N, T + 1, 36, device="cuda", dtype=dt)[:, :-1, :din]
ret = torch.randn(N, T, device="cuda", dtype=dt)
run = critic_ops.make_runner(mlp)
for _ in range(2):
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
ms = best
fl = N * T * 6.0 * (din * H + H * H + H)
print("%s: %.3f ms / epoch (best of 4 x 10) -> %.1f TFLOP/s = %.1f %% of %.1f" % (tag, ms, fl / ms / 1e9, 100 * fl / ms / 1e9 / peak, peak))
