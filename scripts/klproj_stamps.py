"""Per-section cycle stamps of kl_cov_proj_fwd (diagnostic build of gauss.hip
with -DKLP_STAMP, linked into its own .so; the product library is untouched).
    python scripts/klproj_stamps.py [K]"""
import ctypes, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "tce_rl_amd", "csrc")
so = os.path.join(ROOT, "scripts", "variants", "libklp_stamp.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
if "--build" in sys.argv or not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC",
                           "-std=c++17", "-shared", "-DKLP_STAMP", "-DSMJ_STAMP", os.path.join(CS, "gauss.hip"),
                           os.path.join(CS, "capi.hip"), "-o", so])
    if "--build" in sys.argv:
        sys.exit(0)
lib = ctypes.CDLL(so)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
K = int(args[0]) if args else 24
g = torch.Generator().manual_seed(0)
def chol(scale):
    A = torch.randn(K, K, generator=g, dtype=torch.float64) * 0.3
    return torch.linalg.cholesky(A @ A.T + torch.eye(K, dtype=torch.float64) * scale)
Lo = chol(1.0).float().cuda().reshape(1, K, K)
L0 = chol(1.0).float().reshape(1, K, K)
D = 0.002 * torch.tril(torch.randn(1, K, K, generator=g))
lib.tce_kl_cov_proj_ctx_len.restype = ctypes.c_int64
n = lib.tce_kl_cov_proj_ctx_len(K)
f = lib.tce_kl_cov_proj_fwd_f32
f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p,
              ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
              ctypes.c_int, ctypes.c_void_p]
names = ["load", "trsm", "kl0", "jacobi", "eta", "ctx+Y", "mm_nt", "cholesky", "entropy+store"]
for warm in (0, 1):
    ctx = torch.zeros(1, n, dtype=torch.float64, device="cuda")
    out = torch.empty(1, K, K, device="cuda")
    acc = torch.zeros(16, dtype=torch.float64)
    for i in range(20):
        Lk = (L0 + i * D).cuda().contiguous()
        rc = f(Lk.data_ptr(), Lo.data_ptr(), 0, 5e-4, None, 0, out.data_ptr(), ctx.data_ptr(), 1, K, warm, None)
        assert rc == 0
        torch.cuda.synchronize()
        if i >= 2:
            acc += ctx[0, -16:].cpu()
    acc /= 18
    print("K %d warm %d cycles:" % (K, warm), {k: int(v) for k, v in zip(names, acc[:9].tolist())}, "total", int(acc[:9].sum()))
    jn = ["pairs", "loads+dots", "dpp", "rotate+store", "barrier", "sweeps"]
    print("   jacobi (thread 0):", {k: round(v, 1) for k, v in zip(jn, acc[9:15].tolist())})
