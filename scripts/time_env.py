"""Device time of the env rollout kernel at the C2 shape, and of diagnostic
variants with parts switched off (ENV_SKIP bits: 1 state stores, 2 moments,
4 reward stores).  python scripts/time_env.py [--build]"""
import ctypes, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "tce_rl_amd", "csrc")
VAR = os.path.join(ROOT, "scripts", "variants")
os.makedirs(VAR, exist_ok=True)
skips = [0, 1, 2, 4, 7]
if "--build" in sys.argv:
    for sk in skips:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared",
                               "-DENV_SKIP=%d" % sk, os.path.join(CS, "env.hip"), os.path.join(CS, "capi.hip"),
                               "-o", os.path.join(VAR, "libenv%d.so" % sk)])
    sys.exit(0)
N, T, dof, d_task = 4096, 500, 4, 39
D = d_task + 1 + 2 * dof
g = torch.Generator(device="cuda").manual_seed(0)
acts = torch.randn(N, T, 2 * dof, device="cuda", generator=g)
obs0 = torch.randn(N, D, device="cuda", generator=g)
states = torch.empty(N, T + 1, D, device="cuda")
rew = torch.empty(N, T, device="cuda")
met = torch.empty(N, 2, device="cuda")
shift = torch.zeros(D, device="cuda")
part = torch.empty(N, D, 2, dtype=torch.float64, device="cuda")
for sk in skips:
    lib = ctypes.CDLL(os.path.join(VAR, "libenv%d.so" % sk))
    f = lib.tce_env_rollout_f32
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                  ctypes.c_float, ctypes.c_float, ctypes.c_float] + [ctypes.c_void_p] * 7
    run = lambda: f(acts.data_ptr(), obs0.data_ptr(), 0, N, T, dof, d_task, 0.0125, 400.0, 40.0, states.data_ptr(),
                    rew.data_ptr(), None, met.data_ptr(), shift.data_ptr(), part.data_ptr(), None)
    assert run() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    print("ENV_SKIP %d: %.1f us" % (sk, e0.elapsed_time(e1) * 100))
