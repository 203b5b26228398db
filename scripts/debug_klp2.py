import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd import _lib, ops
from tce_rl_amd._lib import call, ptr, stream
from oracle import tce_oracle as O
from oracle import kl_oracle as KO
F64 = torch.float64
K = 63
g = torch.Generator().manual_seed(100 + K)
def rand_chol(K, scale, g, B=1):
    vec = torch.cat([scale * torch.randn(B, K, generator=g, dtype=F64), 0.1 * scale * torch.randn(B, K * (K - 1) // 2, generator=g, dtype=F64)], -1)
    return O.vector_to_cholesky(vec, K, 1e-3, False)
L_o = rand_chol(K, 1.0, g, 3); L = rand_chol(K, 1.0, g, 3)
eps = 5e-3
for b in (0, 1):
    Lb, Lob = L[b:b+1].contiguous(), L_o[b:b+1].contiguous()
    pc, eta_ref = KO.cov_projection(Lb @ Lb.transpose(-1, -2), Lob, eps)
    ref = torch.linalg.cholesky(pc)
    for impl in (1, 0):
        call("tce_kl_proj_impl", impl)
        n = _lib.load().tce_kl_cov_proj_ctx_len(K)
        ctx = torch.zeros(1, n, dtype=F64, device="cuda")
        out = torch.empty(1, K, K, dtype=F64, device="cuda")
        Lg, Log = Lb.cuda(), Lob.cuda()
        call("tce_kl_cov_proj_fwd_f64", ptr(Lg), ptr(Log), 0, eps, None, 0, ptr(out), ptr(ctx), 1, K, 0, stream())
        torch.cuda.synchronize()
        tail = 4 * K * K if impl else K * K + K
        eta = ctx[0, tail].item()
        print("b", b, "impl", impl, "eta %.12g ref %.12g rel %.2e  evals %s  proj err %.2e" % (
            eta, eta_ref.item(), abs(eta - eta_ref.item()) / eta_ref.item(),
            ctx[0, tail + 5].item() if impl else "-", (out.cpu() - ref).abs().max().item()))
        if impl:
            To = ctx[0, :K * K].view(K, K).cpu()
            A = ctx[0, K * K:2 * K * K].view(K, K).cpu()
            print("   To err %.2e  A err %.2e" % ((To - torch.linalg.inv(Lob[0])).abs().max().item() / torch.linalg.inv(Lob[0]).abs().max().item(),
                                                 (A - torch.linalg.solve_triangular(Lob[0], Lb[0], upper=False)).abs().max().item()))
call("tce_kl_proj_impl", 2)
