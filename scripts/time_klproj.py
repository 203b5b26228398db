"""kl_cov_proj_fwd: cold start vs warm start over a drifting sequence (K 24, fp32 I/O)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd import _lib
from tce_rl_amd._lib import call, ptr, stream
K = int(sys.argv[1]) if len(sys.argv) > 1 else 24
g = torch.Generator().manual_seed(0)
def chol(scale):
    A = torch.randn(K, K, generator=g, dtype=torch.float64) * 0.3
    return torch.linalg.cholesky(A @ A.T + torch.eye(K, dtype=torch.float64) * scale)
Lo = chol(1.0).float().cuda().reshape(1, K, K)
L0 = chol(1.0).float().reshape(1, K, K)
D = 0.002 * torch.tril(torch.randn(1, K, K, generator=g))
n = _lib.load().tce_kl_cov_proj_ctx_len(K)
for warm in (0, 1):
    ctx = torch.zeros(1, n, dtype=torch.float64, device="cuda")
    out = torch.empty(1, K, K, device="cuda")
    Ls = [(L0 + i * D).cuda().contiguous() for i in range(50)]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for Lk in Ls:
        call("tce_kl_cov_proj_fwd_f32", ptr(Lk), ptr(Lo), 0, 5e-4, None, 0, ptr(out), ptr(ctx), 1, K, warm, stream())
    e1.record(); torch.cuda.synchronize()
    print("K %d warm_start %d: %.1f us per call" % (K, warm, e0.elapsed_time(e1) * 1e3 / 50))
