"""Kernels of the fused policy objective (csrc/objective.hip) against the CPU
oracle / plain torch: surrogate loss, shared-covariance KL diagnostics,
trust-region loss and its gradients."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _chol(K, g, dtype, scale=0.05):
    diag = torch.nn.functional.softplus(torch.randn(K, generator=g,
                                                    dtype=dtype)) + 1e-2
    L = torch.diag(diag)
    idx = torch.tril_indices(K, K, -1)
    L[idx[0], idx[1]] = scale * torch.randn(idx.shape[1], generator=g,
                                            dtype=dtype)
    return L


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("M", [1, 1000, 4096 * 24])
def test_surrogate(dtype, M):
    from tce_rl_amd import _lib
    from tce_rl_amd._lib import call, ptr, sfx, stream
    g = torch.Generator().manual_seed(M)
    lp_new = torch.randn(M, generator=g, dtype=dtype) * 0.3
    lp_old = lp_new + 0.1 * torch.randn(M, generator=g, dtype=dtype)
    adv = torch.randn(M, generator=g, dtype=dtype)
    x = lp_new.clone().requires_grad_(True)
    ratio = (x - lp_old).exp()
    loss = -(ratio * adv).mean()
    loss.backward()
    out = torch.empty(2, dtype=dtype, device="cuda")
    grad = torch.empty(M, dtype=dtype, device="cuda")
    a, b, c = lp_new.cuda(), lp_old.cuda(), adv.cuda()
    ws = torch.zeros(_lib.load().tce_surrogate_ws_len(), dtype=torch.float64,
                     device="cuda")
    for _ in range(2):                   # the ticket re-arms itself
        out.zero_()
        call("tce_surrogate_" + sfx(dtype), ptr(a), ptr(b), ptr(c), M,
             ptr(out), ptr(grad), ptr(ws), stream())
    tol = 2e-5 if dtype == torch.float32 else 1e-12
    assert out[0].item() == pytest.approx(loss.item(), rel=tol, abs=tol)
    assert out[1].item() == pytest.approx(ratio.mean().item(), rel=tol)
    torch.testing.assert_close(grad.cpu(), x.grad, rtol=tol, atol=tol / M)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("N,K", [(1, 24), (100, 24), (4096, 24), (77, 63),
                                 (300, 8)])
@pytest.mark.parametrize("include_cov", [1, 0])
def test_kl_shared_vs_oracle(dtype, N, K, include_cov):
    from oracle import kl_oracle as KO
    from tce_rl_amd import _lib
    from tce_rl_amd._lib import call, ptr, sfx, stream
    g = torch.Generator().manual_seed(N * 100 + K)
    f64 = torch.float64
    means = [0.3 * torch.randn(N, K, generator=g, dtype=f64) for _ in range(3)]
    Ls = [_chol(K, g, f64) for _ in range(3)]
    mn, mo, mp = means
    Ln, Lo, Lp = Ls
    coeff = 1.7
    # reference in float64 on the CPU
    mn_r, Ln_r = mn.clone().requires_grad_(True), Ln.clone().requires_grad_(True)
    ex = lambda L: L.unsqueeze(0).expand(N, -1, -1)
    want = []
    for (ma, La), (mb, Lb) in (((mn_r, Ln_r), (mo, Lo)), ((mn_r, Ln_r), (mp, Lp)),
                               ((mp, Lp), (mo, Lo))):
        want.extend(t.mean() for t in
                    KO.gaussian_kl_details(ma, ex(La), mb, ex(Lb)))
    maha, cov = KO.gaussian_kl(mn_r, ex(Ln_r), mp, ex(Lp))
    tr = coeff * ((maha + cov).mean() if include_cov else maha.mean())
    tr.backward()
    ent = 0.5 * K * (1 + math.log(2 * math.pi)) + Lp.diagonal().log().sum()
    # kernel
    d = lambda t: t.to(dtype).cuda().contiguous()
    out = torch.empty(16, dtype=dtype, device="cuda")
    gm = torch.empty(N, K, dtype=dtype, device="cuda")
    gL = torch.empty(K, K, dtype=dtype, device="cuda")
    ws = torch.empty(_lib.load().tce_kl_shared_ws_len(N), dtype=f64,
                     device="cuda")
    args = [d(t) for t in (mn, mo, mp, Ln, Lo, Lp)]
    call("tce_kl_shared_" + sfx(dtype), *[ptr(t) for t in args], N, K, coeff,
         include_cov, ptr(out), ptr(gm), ptr(gL), ptr(ws), stream())
    tol = 3e-4 if dtype == torch.float32 else 1e-9
    got = out.double().cpu()
    for i, w in enumerate(want):
        assert got[i].item() == pytest.approx(w.item(), rel=tol, abs=tol), i
    assert got[12].item() == pytest.approx(ent.item(), rel=tol)
    assert got[13].item() == pytest.approx(tr.item(), rel=tol, abs=tol)
    torch.testing.assert_close(gm.double().cpu(), mn_r.grad, rtol=tol,
                               atol=tol * mn_r.grad.abs().max().item())
    want_gL = torch.tril(Ln_r.grad) if include_cov else torch.zeros(K, K,
                                                                   dtype=f64)
    torch.testing.assert_close(gL.double().cpu(), want_gL, rtol=tol,
                               atol=tol * max(want_gL.abs().max().item(), 1.0))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("N,K,H", [(1, 24, 128), (33, 8, 64), (4096, 24, 128),
                                   (1000, 63, 256), (257, 3, 128)])
def test_out_layer_grad(dtype, N, K, H):
    """dW = g^T h, db = sum g of a final Linear layer against torch float64."""
    from tce_rl_amd import _lib
    from tce_rl_amd._lib import call, ptr, sfx, stream
    gen = torch.Generator().manual_seed(N + K + H)
    g = torch.randn(N, K, generator=gen, dtype=torch.float64)
    h = torch.randn(N, H, generator=gen, dtype=torch.float64)
    gd, hd = g.to(dtype).cuda(), h.to(dtype).cuda()
    dW = torch.empty(K, H, dtype=dtype, device="cuda")
    db = torch.empty(K, dtype=dtype, device="cuda")
    ws = torch.empty(_lib.load().tce_out_layer_grad_ws_len(N, K, H),
                     dtype=dtype, device="cuda")
    call("tce_out_layer_grad_" + sfx(dtype), ptr(gd), ptr(hd), ptr(dW),
         ptr(db), ptr(ws), N, K, H, stream())
    tol = 2e-5 if dtype == torch.float32 else 1e-12
    scale = math.sqrt(N)
    torch.testing.assert_close(dW.double().cpu(), g.t() @ h, rtol=tol,
                               atol=tol * scale)
    torch.testing.assert_close(db.double().cpu(), g.sum(0), rtol=tol,
                               atol=tol * scale)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("ent_coef", [0.0, 0.05])
def test_policy_record_row(dtype, ent_coef):
    from tce_rl_amd._lib import call, ptr, sfx, stream
    sur = torch.tensor([0.7, 1.01], dtype=dtype, device="cuda")
    out = torch.arange(16, dtype=dtype, device="cuda") * 0.1 + 1
    norms = torch.tensor([3.0, 0.5], dtype=dtype, device="cuda")
    row = torch.empty(19, dtype=dtype, device="cuda")
    call("tce_policy_record_" + sfx(dtype), ptr(sur), ptr(out), ptr(norms),
         ent_coef, ptr(row), stream())
    entl = -ent_coef * out[12]
    want = torch.cat([sur[:1], entl[None], out[13:14],
                      (sur[0] + out[13] + entl)[None], out[12:13], norms,
                      out[:12]])
    torch.testing.assert_close(row, want, rtol=1e-6, atol=0)


def test_objective_entry_points_reject_bad_arguments():
    import ctypes
    lib = __import__("tce_rl_amd._lib", fromlist=["load"]).load()
    x = torch.zeros(64, device="cuda")
    p = x.data_ptr()
    assert lib.tce_out_layer_grad_f32(p, p, p, p, p, 4, 65, 128, None) != 0
    assert b"K <= 64" in lib.tce_last_error()
    assert lib.tce_out_layer_grad_f32(None, p, p, p, p, 4, 8, 128, None) != 0
    assert lib.tce_policy_record_f32(None, p, p, 0.0, p, None) != 0
    assert lib.tce_policy_objective_end_f32(None, p, 4, 8, 2, None) != 0
    assert lib.tce_policy_objective_begin_f32(None, 3, 1e-3, p, 1e-3, None, 0,
                                              p, p, p, 4, 8, 2, None) != 0
    assert lib.tce_bb_policy_objective_f32(
        p, p, p, p, None, p, p, 0.01, 1e-3, None, 0, p, 1.0, 1, 0.0, p, p, p,
        p, p, p, p, None, None, 4, 8, None) != 0
    assert b"null buffer" in lib.tce_last_error()
    # (no [N, K, K] tensor: the shared factor's gradient is one product over the envs)
    assert 4 * 4096 * 20 < lib.tce_bb_policy_objective_ws_len(4096, 20) < 4096 * 400


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("N,K", [(37, 5), (300, 28), (4096, 63)])
def test_logprob_backward_with_z_matches_autograd(N, K, dtype):
    """tce_mvn_logprob_bwd_z_* / _sur_* (black_box_policy.py:95-128 under a
    factor shared by all envs): the per-env gradient w.r.t. the mean, the
    log-probs, and -- from z -- the factor's gradient as ONE product over the
    envs, tril((g q)^T z) - (sum g) diag(1 / L_ii), against torch autograd of
    the surrogate loss in float64."""
    from tce_rl_amd._lib import call, ptr, sfx, stream
    g = torch.Generator().manual_seed(N + K)
    d64 = torch.float64
    L = torch.tril(torch.randn(K, K, generator=g, dtype=d64)) * 0.2
    L.diagonal().copy_(torch.rand(K, generator=g, dtype=d64) + 0.5)
    mean = torch.randn(N, K, generator=g, dtype=d64)
    x = mean + (torch.randn(N, K, generator=g, dtype=d64) @ L.T)
    adv = torch.randn(N, generator=g, dtype=d64)
    m_ref = mean.clone().requires_grad_(True)
    L_ref = L.clone().requires_grad_(True)
    dist = torch.distributions.MultivariateNormal(m_ref, scale_tril=L_ref)
    logp = dist.log_prob(x)
    lp_old = (logp.detach() + 0.1 * torch.randn(N, generator=g, dtype=d64))
    loss = -((logp - lp_old).exp() * adv).mean()
    loss.backward()
    dev = lambda t: t.to(dtype).cuda().contiguous()
    xd, md, Ld, lod, ad = dev(x), dev(mean), dev(L), dev(lp_old), dev(adv)
    gm = torch.empty(N, K, dtype=dtype, device="cuda")
    z = torch.empty(N, K, dtype=dtype, device="cuda")
    lp = torch.empty(N, dtype=dtype, device="cuda")
    call("tce_mvn_logprob_bwd_z_sur_" + sfx(dtype), ptr(xd), ptr(md), ptr(Ld),
         ptr(lod), ptr(ad), ptr(gm), ptr(z), ptr(lp), N, K, stream())
    tol = dict(rtol=2e-4, atol=2e-5) if dtype == torch.float32 else \
        dict(rtol=1e-9, atol=1e-11)
    torch.testing.assert_close(lp.cpu().double(), logp.detach(), **tol)
    torch.testing.assert_close(gm.cpu().double(), m_ref.grad, **tol)
    glp = -((logp.detach() - lp_old).exp() * adv) / N
    gL = torch.tril(gm.cpu().double().T @ z.cpu().double()) - \
        glp.sum() * torch.diag(1.0 / L.diagonal())
    torch.testing.assert_close(gL, torch.tril(L_ref.grad), **tol)
    # the variant that takes d loss / d logp from the caller
    gm2 = torch.empty_like(gm)
    z2 = torch.empty_like(z)
    call("tce_mvn_logprob_bwd_z_" + sfx(dtype), ptr(xd), ptr(md), ptr(Ld),
         ptr(dev(glp)), ptr(gm2), ptr(z2), N, K, stream())
    torch.testing.assert_close(gm2.cpu().double(), m_ref.grad, **tol)
    torch.testing.assert_close(z2, z)
