"""kl_cov_proj fwd / bwd per call over a drifting sequence (one policy update):
both implementations (tce_kl_proj_impl 1 = Newton / Gauss-Jordan on the MFMA,
0 = Jacobi), K 24 and 63, projection active in every call.
    python scripts/time_klproj2.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd import _lib
from tce_rl_amd._lib import call, ptr, stream
from oracle import tce_oracle as O
torch.manual_seed(0)
for K in (24, 63):
    g = torch.Generator().manual_seed(K)
    def rc(scale):
        vec = torch.cat([scale * torch.randn(1, K, generator=g, dtype=torch.float64),
                         0.1 * scale * torch.randn(1, K * (K - 1) // 2, generator=g, dtype=torch.float64)], -1)
        return O.vector_to_cholesky(vec, K, 1e-3, False)
    Lo = rc(1.0).cuda()
    # a drift that takes the unprojected KL from 0 to ~2e-2 over the 50 calls
    # (bound 5e-4: what 50 policy epochs do)
    D = (6e-3 / K) * torch.tril(torch.randn(1, K, K, generator=g, dtype=torch.float64)).cuda()
    W = torch.randn(1, K, K, generator=g, dtype=torch.float64).cuda()
    for dt in (torch.float32, torch.float64):
        for impl in (1, 0):
            call("tce_kl_proj_impl", impl)
            n = _lib.load().tce_kl_cov_proj_ctx_len(K)
            sfx = "f32" if dt == torch.float32 else "f64"
            for warm in (1, 0):
                ctx = torch.zeros(1, n, dtype=torch.float64, device="cuda")
                Ls = [(Lo + (1 + s) * D).to(dt).contiguous() for s in range(50)]
                Lod, Wd = Lo.to(dt).contiguous(), W.to(dt).contiguous()
                out, gl = torch.empty_like(Ls[0]), torch.empty_like(Ls[0])
                ef = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                tf = tb = 0.0
                act = 0
                evs = 0
                for Lk in Ls:
                    ef[0].record()
                    call("tce_kl_cov_proj_fwd_" + sfx, ptr(Lk), ptr(Lod), 0, 5e-4, None, 0, ptr(out), ptr(ctx), 1, K, warm, stream())
                    ef[1].record()
                    call("tce_kl_cov_proj_bwd_" + sfx, ptr(Lk), ptr(Lod), 0, ptr(out), ptr(ctx), ptr(Wd), ptr(gl), 1, K, stream())
                    ef[2].record()
                    torch.cuda.synchronize()
                    tf += ef[0].elapsed_time(ef[1]); tb += ef[1].elapsed_time(ef[2])
                    tail = 4 * K * K if impl else K * K + K
                    act += int(ctx[0, tail + 1].item())
                    if impl and ctx[0, tail + 1].item():
                        evs += int(ctx[0, tail + 5].item())
                print("K %d %s impl %s warm %d: fwd %.1f us  bwd %.1f us  (active %d/50, %d evaluations of h)" % (
                    K, sfx, "newton" if impl else "jacobi", warm, tf / 50 * 1e3, tb / 50 * 1e3, act, evs))
call("tce_kl_proj_impl", 2)
