"""Cycles per phase of mlpw_chain_kernel (diagnostic build with -DMLPW_STAMP in
its own .so) at the C3 critic shape.  python scripts/mlpw_stamps.py [--build] [f64]"""
import ctypes, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "tce_rl_amd", "csrc")
so = os.path.join(ROOT, "scripts", "variants", "libmlpw_stamp.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
if "--build" in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared",
                           "-DMLPW_STAMP", "-DMLPW_ONLY_RELU", os.path.join(CS, "mlpw_f32.hip"),
                           os.path.join(CS, "mlpw_f64.hip"), os.path.join(CS, "capi.hip"), "-o", so])
    sys.exit(0)
f64 = "f64" in sys.argv
dt = torch.float64 if f64 else torch.float32
lib = ctypes.CDLL(so)
N, T, din, H = 8192, 100, 21, 256
R = N * T
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(R, din, device="cuda", generator=g, dtype=dt)
ret = torch.randn(R, device="cuda", generator=g, dtype=dt)
w1 = torch.randn(H, din, device="cuda", generator=g, dtype=dt) * 0.2
w2 = torch.randn(H, H, device="cuda", generator=g, dtype=dt) * 0.06
w3 = torch.randn(H, device="cuda", generator=g, dtype=dt) * 0.06
b = torch.zeros(H, device="cuda", dtype=dt)
lib.tce_mlpw_workspace_len.restype = ctypes.c_int64
lib.tce_mlpw_num_params.restype = ctypes.c_int64
ws = torch.empty(lib.tce_mlpw_workspace_len(ctypes.c_int64(R), H, 1), device="cuda", dtype=dt)
P = lib.tce_mlpw_num_params(din, H)
part = torch.empty(lib.tce_mlpw_grid(), P + 2, device="cuda", dtype=dt)
grad = torch.empty(P, device="cuda", dtype=dt)
stats = torch.zeros(2, device="cuda", dtype=dt)
fn = lib.tce_mlpw_critic_f64 if f64 else lib.tce_mlpw_critic_f32
rl = ctypes.c_double if f64 else ctypes.c_float
fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int] + \
    [ctypes.c_void_p] * 6 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, rl] + [ctypes.c_void_p] * 5 + [ctypes.c_int] + \
    [ctypes.c_void_p] * 4 + [rl] * 7 + [ctypes.c_void_p] * 2
p = lambda t: t.data_ptr()
for it in range(3):
    rc = fn(p(x), 0, din, R, R, din, H, p(w1), p(b), p(w2), p(b), p(w3), p(b), 1, p(ret), None, 0.0, None, p(ws), p(part),
            p(grad), p(stats), 0, None, None, None, None, 0, 0, 0, 0, 0, 0, 1, None, None)
    assert rc == 0
    torch.cuda.synchronize()
dy1 = ws[2 * H * H + 2 * R * H:]
st = dy1[:8].cpu().tolist()
names = ["layer1+h1 store", "fwd: barrier", "loss+dY2", "bwd panels", "fwd: fetch+mma", "fwd: stash", "fwd: epilogue", "tile head"]
tiles = (R + 127) // 128 / lib.tce_mlpw_grid()
print("tiles per workgroup %.1f; cycles per tile:" % tiles, {n: int(v / tiles) for n, v in zip(names, st) if n != "-"},
      "total/tile", int(sum(st) / tiles))
