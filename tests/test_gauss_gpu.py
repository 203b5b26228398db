"""GPU parity of the Gaussian-head / KL trust-region kernels vs the oracle."""
import numpy as np
import pytest
import torch

from oracle import kl_oracle as KO
from oracle import tce_oracle as O

pytestmark = pytest.mark.gpu
T_ = torch.as_tensor
F64 = torch.float64


@pytest.fixture(scope="module")
def ops():
    from tce_rl_amd import ops
    return ops


@pytest.fixture(params=["newton", "jacobi"])
def klp_impl(request):
    """Both forms of the covariance projection kernels: "newton" = without an
    eigen-decomposition (csrc/klproj2.h, the default), "jacobi" = rounds 1 - 3."""
    from tce_rl_amd._lib import call
    call("tce_kl_proj_impl", int(request.param == "newton"))
    yield request.param
    call("tce_kl_proj_impl", 2)


def rand_chol(K, scale, g, B=1, dtype=F64):
    vec = torch.cat([scale * torch.randn(B, K, generator=g, dtype=dtype),
                     0.1 * scale * torch.randn(B, K * (K - 1) // 2, generator=g,
                                               dtype=dtype)], -1)
    return O.vector_to_cholesky(vec, K, 1e-3, False)


def test_chol_build_golden_and_backward(ops, golden):
    g = golden("cholesky_head")
    for K in (20, 24, 28, 36, 63):
        for std_only in (False, True):
            tag = f"K{K}_{'diag' if std_only else 'full'}"
            vec = T_(g[f"vec_{tag}"])
            L = ops.chol_build(vec.cuda(), K, 1e-5)
            np.testing.assert_allclose(L.cpu().numpy(), g[f"L_{tag}"],
                                       rtol=1e-6, atol=1e-7)
    vec = torch.randn(3, 36 + 36 * 35 // 2, dtype=F64)
    vec[0, 0] = 25.0                       # softplus threshold branch
    W = torch.randn(3, 36, 36, dtype=F64)
    v_c = vec.clone().requires_grad_(True)
    (O.vector_to_cholesky(v_c, 36, 1e-4, False) * W).sum().backward()
    v_g = vec.cuda().requires_grad_(True)
    (ops.chol_build(v_g, 36, 1e-4) * W.cuda()).sum().backward()
    torch.testing.assert_close(v_g.grad.cpu(), v_c.grad, rtol=1e-12, atol=1e-12)


def test_mvn_golden(ops, golden):
    g = golden("mvn")
    for K in (20, 36):
        mean = T_(g[f"mean_K{K}"]).cuda().requires_grad_(True)
        L = T_(g[f"L_K{K}"]).cuda().requires_grad_(True)
        x = ops.mvn_rsample(mean.detach(), L.detach(), T_(g[f"eps_K{K}"]).cuda())
        np.testing.assert_allclose(x.cpu().numpy(), g[f"x_K{K}"], rtol=1e-5,
                                   atol=1e-5)
        lp = ops.mvn_log_prob(T_(g[f"x_K{K}"]).cuda(), mean, L)
        np.testing.assert_allclose(lp.detach().cpu().numpy(), g[f"logp_K{K}"],
                                   rtol=1e-5, atol=1e-4)
        (lp * T_(g[f"w_K{K}"]).cuda()).sum().backward()
        # fp32 triangular solves with small pivots (diag ~ softplus + 1e-5):
        # gradients agree to fp32 conditioning; exactness is checked in fp64
        # by test_mvn_logprob_backward_f64
        np.testing.assert_allclose(mean.grad.cpu().numpy(), g[f"dmean_K{K}"],
                                   rtol=2e-3, atol=5e-3)
        np.testing.assert_allclose(np.tril(L.grad.cpu().numpy()),
                                   g[f"dL_K{K}"], rtol=2e-3, atol=5e-3)
        np.testing.assert_allclose(
            ops.mvn_entropy(L.detach()).cpu().numpy(), g[f"ent_K{K}"], rtol=1e-6)
        np.testing.assert_allclose(
            ops.log_determinant(L.detach()).cpu().numpy(), g[f"logdet_K{K}"],
            rtol=1e-6)
        np.testing.assert_allclose(
            ops.maha(mean.detach(), T_(g[f"other_K{K}"]).cuda(),
                     L.detach()).cpu().numpy(), g[f"maha_K{K}"], rtol=1e-5)


@pytest.mark.parametrize("shared", [True, False])
def test_mvn_logprob_backward_f64(ops, shared):
    g = torch.Generator().manual_seed(0)
    N, K = 9, 20
    L = rand_chol(K, 1.0, g, 1 if shared else N)
    mean = torch.randn(N, K, generator=g, dtype=F64)
    x = mean + torch.randn(N, K, generator=g, dtype=F64)
    w = torch.randn(N, generator=g, dtype=F64)
    m_c = mean.clone().requires_grad_(True)
    L_c = L.clone().requires_grad_(True)
    (O.mvn_log_prob(x, m_c, L_c.expand(N, -1, -1)) * w).sum().backward()
    m_g = mean.cuda().requires_grad_(True)
    Lb = L.cuda().requires_grad_(True)
    L_g = ops.expand_shared(Lb[0], N) if shared else Lb
    lp = ops.mvn_log_prob(x.cuda(), m_g, L_g)
    torch.testing.assert_close(
        lp.cpu(), O.mvn_log_prob(x, mean, L.expand(N, -1, -1)),
        rtol=1e-10, atol=1e-10)
    (lp * w.cuda()).sum().backward()
    torch.testing.assert_close(m_g.grad.cpu(), m_c.grad, rtol=1e-9, atol=1e-9)
    torch.testing.assert_close(torch.tril(Lb.grad.cpu()), torch.tril(L_c.grad),
                               rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("shared", [True, False])
def test_mean_projection_and_maha(ops, shared):
    g = torch.Generator().manual_seed(1)
    N, K = 33, 36
    L_o = rand_chol(K, 1.0, g, 1 if shared else N)
    mu_o = torch.randn(N, K, generator=g, dtype=F64)
    mu = mu_o + 0.3 * torch.randn(N, K, generator=g, dtype=F64) * \
        torch.rand(N, 1, generator=g, dtype=F64)
    mu[0] = mu_o[0] + 1e-4                  # inactive row
    eps = 0.5
    w = torch.randn(N, K, generator=g, dtype=F64)
    Lo_full = L_o.expand(N, -1, -1)
    mu_c = mu.clone().requires_grad_(True)
    maha_c, _ = KO.gaussian_kl(mu_c, Lo_full, mu_o, Lo_full)
    pm_c = KO.mean_projection(mu_c, mu_o, maha_c, eps)
    (pm_c * w).sum().backward()
    Lg = ops.expand_shared(L_o[0].cuda(), N) if shared else L_o.cuda()
    mu_g = mu.cuda().requires_grad_(True)
    pm = ops.kl_mean_projection(mu_g, mu_o.cuda(), Lg, eps)
    torch.testing.assert_close(pm.cpu(), pm_c.detach(), rtol=1e-10, atol=1e-10)
    (pm * w.cuda()).sum().backward()
    torch.testing.assert_close(mu_g.grad.cpu(), mu_c.grad, rtol=1e-8, atol=1e-9)
    # tightness: projected mean part == eps where active
    m2 = 0.5 * ops.maha(pm.detach(), mu_o.cuda(), Lg).cpu()
    active = maha_c.detach() > eps
    assert active.any() and (~active).any()
    torch.testing.assert_close(m2[active], torch.full_like(m2[active], eps),
                               rtol=1e-9, atol=1e-12)
    # maha backward
    x_g = mu.cuda().requires_grad_(True)
    (ops.maha(x_g, mu_o.cuda(), Lg) * w[:, 0].cuda()).sum().backward()
    x_c = mu.clone().requires_grad_(True)
    (O.maha(x_c, mu_o, Lo_full) * w[:, 0]).sum().backward()
    torch.testing.assert_close(x_g.grad.cpu(), x_c.grad, rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("shared", [True, False])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_mean_projection_backward_accumulating_form(ops, shared, dtype):
    """tce_mean_proj_bwd_acc_* == tce_vec_env_*(mode 1, bwd) + an add, bit for
    bit (the policy objective's join of the two halves of d / d mean)."""
    from tce_rl_amd._lib import call, ptr, sfx, stream
    g = torch.Generator().manual_seed(5)
    N, K = 70, 24
    L_o = rand_chol(K, 1.0, g, 1 if shared else N).to(dtype)
    mu_o = torch.randn(N, K, generator=g, dtype=F64)
    mu = (mu_o + 0.3 * torch.randn(N, K, generator=g, dtype=F64)
          * torch.rand(N, 1, generator=g, dtype=F64)).to(dtype).cuda()
    mu_o = mu_o.to(dtype).cuda()
    go = torch.randn(N, K, generator=g, dtype=F64).to(dtype).cuda()
    base = torch.randn(N, K, generator=g, dtype=F64).to(dtype).cuda()
    Lg = L_o[0].cuda().contiguous() if shared else L_o.cuda().contiguous()
    sL = 0 if shared else K * K
    gx = torch.empty_like(mu)
    call("tce_vec_env_" + sfx(dtype), 1, 1, ptr(mu), ptr(mu_o), ptr(Lg), sL, 0.5,
         ptr(go), None, ptr(gx), None, N, K, stream())
    acc = base.clone()
    call("tce_mean_proj_bwd_acc_" + sfx(dtype), ptr(mu), ptr(mu_o), ptr(Lg), sL,
         0.5, ptr(go), None, ptr(acc), N, K, stream())
    assert torch.equal(acc, base + gx)
    # ... and with z = L^-1 (x - y) handed over by the forward pass (shared L),
    # which also returns |z|^2 per env
    if shared:
        pm, quad, z = (torch.empty_like(mu), torch.empty(N, dtype=dtype, device="cuda"),
                       torch.empty_like(mu))
        call("tce_mean_proj_fwd_q_" + sfx(dtype), ptr(mu), ptr(mu_o), ptr(Lg), sL,
             0.5, ptr(pm), ptr(quad), ptr(z), N, K, stream())
        ref = torch.empty_like(mu)
        call("tce_vec_env_" + sfx(dtype), 1, 0, ptr(mu), ptr(mu_o), ptr(Lg), sL, 0.5,
             None, ptr(ref), None, None, N, K, stream())
        assert torch.equal(pm, ref)
        torch.testing.assert_close(quad, (z * z).sum(-1), rtol=1e-5 if dtype == torch.float32 else 1e-12, atol=0)
        acc2 = base.clone()
        call("tce_mean_proj_bwd_acc_" + sfx(dtype), ptr(mu), ptr(mu_o), ptr(Lg), sL,
             0.5, ptr(go), ptr(z), ptr(acc2), N, K, stream())
        assert torch.equal(acc2, acc)


@pytest.mark.parametrize("K", [5, 24, 36, 63])
def test_kl_cov_part_fwd_bwd(ops, K):
    g = torch.Generator().manual_seed(K)
    B = 3
    L = rand_chol(K, 1.0, g, B)
    L_o = rand_chol(K, 1.0, g, B)
    w = torch.randn(B, generator=g, dtype=F64)
    L_c = L.clone().requires_grad_(True)
    z = torch.zeros(B, K, dtype=F64)
    _, cp = KO.gaussian_kl(z, L_c, z, L_o)
    (cp * w).sum().backward()
    L_g = L.cuda().requires_grad_(True)
    out = ops.kl_cov_part(L_g, L_o.cuda(), B)
    torch.testing.assert_close(out.cpu(), cp.detach(), rtol=1e-10, atol=1e-11)
    (out * w.cuda()).sum().backward()
    torch.testing.assert_close(torch.tril(L_g.grad.cpu()), torch.tril(L_c.grad),
                               rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("K", [6, 24, 36, 63])
@pytest.mark.parametrize("use_beta", [False, True])
def test_kl_cov_projection_fwd_bwd(ops, K, use_beta, klp_impl):
    g = torch.Generator().manual_seed(100 + K)
    B = 3
    L_o = rand_chol(K, 1.0, g, B)
    L = rand_chol(K, 1.0, g, B)
    L[2] = L_o[2] * 1.00001                 # inactive matrix
    eps = 5e-3
    W = torch.randn(B, K, K, generator=g, dtype=F64)
    beta = None
    if use_beta:
        beta = torch.tensor(float(KO.entropy(L_o).mean()) + 0.05, dtype=F64)

    def oracle(Lx):
        cov = Lx @ Lx.transpose(-1, -2)
        pc, eta = KO.cov_projection(cov, L_o, eps)
        pl = torch.linalg.cholesky(pc)
        if beta is not None:
            _, pl = KO.entropy_projection(None, pl, beta)
        return pl, eta

    L_c = L.clone().requires_grad_(True)
    pl_c, eta = oracle(L_c)
    assert eta[0] > 0 and eta[1] > 0 and eta[2] == 0
    (pl_c * W).sum().backward()
    L_g = L.cuda().requires_grad_(True)
    pl = ops.kl_cov_projection(L_g, L_o.cuda(), eps,
                               None if beta is None else beta.cuda())
    # (newton: the dual variable is solved until the constraint holds to 1e-10
    # relative, and these matrices are independent draws -- cond(Lo^-1 Sigma
    # Lo^-T) ~ 1e6, far from anything a policy update produces -- where the
    # pivot-free elimination keeps ~1e-9; near the old covariance both forms
    # agree to 1e-11, test_kl_cov_projection_forms_agree_near_the_old_policy)
    tol = dict(rtol=1e-7, atol=1e-8) if klp_impl == "newton" else \
        dict(rtol=1e-8, atol=1e-9)
    torch.testing.assert_close(pl.cpu(), pl_c.detach(), **tol)
    (pl * W.cuda()).sum().backward()
    torch.testing.assert_close(torch.tril(L_g.grad.cpu()), torch.tril(L_c.grad),
                               rtol=1e-6, atol=1e-7)
    # KKT: the projected covariance part of the KL sits on the bound
    if beta is None:
        kl = ops.kl_cov_part(pl.detach(), L_o.cuda(), B).cpu()
        torch.testing.assert_close(kl[:2], torch.full_like(kl[:2], eps),
                                   rtol=1e-7, atol=1e-10)
        assert kl[2] < eps


@pytest.mark.parametrize("K", [24, 36, 63])
def test_kl_cov_projection_forms_agree_near_the_old_policy(ops, K):
    """What a policy update produces: the new factor a small step away from the
    old one, the projection just active.  The eigen-free kernels and the Jacobi
    kernels agree to 1e-10 (forward) / 1e-8 (backward) there."""
    from tce_rl_amd._lib import call
    g = torch.Generator().manual_seed(300 + K)
    L_o = rand_chol(K, 1.0, g, 1)
    out = []
    for scale in (0.004, 0.02):
        L = (L_o + scale * torch.tril(torch.randn(1, K, K, generator=g, dtype=F64)))
        W = torch.randn(1, K, K, generator=g, dtype=F64).cuda()
        res = []
        try:
            for impl in (1, 0):
                call("tce_kl_proj_impl", impl)
                Lg = L.cuda().requires_grad_(True)
                pl = ops.kl_cov_projection(Lg, L_o.cuda(), 5e-4)
                (pl * W).sum().backward()
                res.append((pl.detach(), torch.tril(Lg.grad)))
        finally:
            call("tce_kl_proj_impl", 2)
        assert (res[0][0] - L.cuda()).abs().max() > 1e-6       # the projection was active
        torch.testing.assert_close(res[0][0], res[1][0], rtol=1e-10, atol=1e-10)
        torch.testing.assert_close(res[0][1], res[1][1], rtol=1e-7, atol=1e-8)


def test_kl_cov_projection_fp32_close(ops, klp_impl):
    """fp32 I/O (the Metaworld config): the solve itself is in double like the
    reference's C++ solver, so the result is within fp32 rounding of the fp64
    oracle."""
    g = torch.Generator().manual_seed(7)
    K = 36
    L_o = rand_chol(K, 1.0, g)
    L = rand_chol(K, 1.0, g)
    pc, _ = KO.cov_projection(L @ L.transpose(-1, -2), L_o, 5e-4)
    ref = torch.linalg.cholesky(pc)
    out = ops.kl_cov_projection(L.float().cuda(), L_o.float().cuda(), 5e-4)
    torch.testing.assert_close(out.cpu().double(), ref, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("K", [24, 63])
def test_kl_cov_projection_warm_start(ops, K, klp_impl):
    """warm_start: a chain of 12 slowly drifting covariances (the policy epochs
    of one update), each projection started from what the previous call left in
    the context buffer (eigenvectors / the dual variable and Lo^-1) == the cold
    projection and its backward, to double-precision rounding."""
    from tce_rl_amd import _lib
    from tce_rl_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(200 + K)
    L_o = rand_chol(K, 1.0, g, 1)
    L = rand_chol(K, 1.0, g, 1)
    D = 0.01 * torch.tril(torch.randn(1, K, K, generator=g, dtype=F64))
    W = torch.randn(1, K, K, generator=g, dtype=F64).cuda()
    n = _lib.load().tce_kl_cov_proj_ctx_len(K)
    ctx = torch.zeros(1, n, dtype=F64, device="cuda")
    Lo_g = L_o.cuda()
    for step in range(12):
        Lk = (L + step * D).cuda().contiguous()
        warm = torch.empty_like(Lk)
        call("tce_kl_cov_proj_fwd_f64", ptr(Lk), ptr(Lo_g), 0, 5e-3, None, 0,
             ptr(warm), ptr(ctx), 1, K, 1, stream())
        gw = torch.empty_like(Lk)
        call("tce_kl_cov_proj_bwd_f64", ptr(Lk), ptr(Lo_g), 0, ptr(warm),
             ptr(ctx), ptr(W), ptr(gw), 1, K, stream())
        Lr = Lk.clone().requires_grad_(True)
        cold = ops.kl_cov_projection(Lr, Lo_g, 5e-3)
        (cold * W).sum().backward()
        tail = 4 * K * K if klp_impl == "newton" else K * K + K
        assert ctx[0, tail + 1].item() == 1.0               # projection active
        # (newton: the dual variable is accepted once its Newton step falls
        # under 3e-11 relative -- the noise of h(eta) -- so two different
        # starting points agree to that, not to the last bit)
        tol = dict(rtol=1e-9, atol=1e-10) if klp_impl == "newton" else \
            dict(rtol=1e-10, atol=1e-11)
        torch.testing.assert_close(warm, cold.detach(), **tol)
        torch.testing.assert_close(torch.tril(gw), torch.tril(Lr.grad),
                                   rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("K", [24, 36])
def test_kl_cov_projection_says_so_when_the_dual_search_fails(ops, K):
    """a factor that is not finite makes every comparison of the
    eigen-free form's search for the dual variable false; it used to leave
    after 60 evaluations with whatever it held and a context that looked valid.
    Now the projection and the stored dual variable are NaN -- what the
    agent's NaN check on the losses turns into "NAN ... detected", as the
    reference does (temporal_correlated_agent.py:569-577)."""
    from tce_rl_amd import _lib
    from tce_rl_amd._lib import call, ptr, stream
    lib = _lib.load()
    lib.tce_kl_proj_impl(1)                          # the eigen-free form
    try:
        g = torch.Generator().manual_seed(9)
        L_o = rand_chol(K, 1.0, g, 1).cuda()
        L = rand_chol(K, 1.0, g, 1)                  # an independent draw: the bound is active
        ctx = torch.zeros(1, lib.tce_kl_cov_proj_ctx_len(K), dtype=F64,
                          device="cuda")
        good = torch.empty(1, K, K, dtype=F64, device="cuda")
        call("tce_kl_cov_proj_fwd_f64", ptr(L.cuda()), ptr(L_o), 0, 5e-3, None,
             0, ptr(good), ptr(ctx), 1, K, 0, stream())
        assert torch.isfinite(good).all() and ctx[0, 4 * K * K + 1] == 1.0
        # (a) an old factor so close to singular that the whitened covariance
        # overflows: kl0 = inf, the bound is "active", every evaluation of h is
        # NaN -- the search cannot end
        bad_Lo = L_o.clone()
        bad_Lo[0, K // 2, K // 2] = 1e-200
        out = torch.zeros(1, K, K, dtype=F64, device="cuda")
        ctx.zero_()
        call("tce_kl_cov_proj_fwd_f64", ptr(L.cuda()), ptr(bad_Lo), 0, 5e-3,
             None, 0, ptr(out), ptr(ctx), 1, K, 0, stream())
        torch.cuda.synchronize()
        low = torch.tril(torch.ones(K, K, device="cuda")).bool()
        assert torch.isnan(out[0][low]).all()
        assert torch.isnan(ctx[0, 4 * K * K])            # the stored dual variable
        # (b) a non-finite entry: kl0 is NaN, the factor passes through as it
        # is -- not finite either way
        bad_L = L.clone()
        bad_L[0, K // 2, 1] = float("inf")
        ctx.zero_()
        call("tce_kl_cov_proj_fwd_f64", ptr(bad_L.cuda()), ptr(L_o), 0, 5e-3,
             None, 0, ptr(out), ptr(ctx), 1, K, 0, stream())
        torch.cuda.synchronize()
        assert not torch.isfinite(out).all()
    finally:
        lib.tce_kl_proj_impl(2)                      # the default choice by K


@pytest.mark.parametrize("K", [4, 12, 24])
def test_kl_cov_projection_kernel_is_the_constrained_optimum(ops, K, klp_impl):
    """The projection KERNEL (either form) against a direct SLSQP solution of
        min KL_cov(S~ || S)  s.t.  KL_cov(S~ || S_old) <= eps
    (tests/test_kl_optimum_cpu.py; nothing of oracle/kl_oracle.py involved)."""
    import os
    import numpy as np
    from test_kl_optimum_cpu import kl_cov
    # the SLSQP solutions come from tests/golden/kl_slsqp.npz (generator:
    # tests/golden/make_kl_slsqp.py; one K = 24 solve takes minutes on a GPU
    # box's host share -- tests/test_kl_optimum_cpu.py re-solves the small
    # cases against the file)
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "kl_slsqp.npz"))
    S, S_old, eps = gold["S_%d" % K], gold["S_old_%d" % K], float(gold["eps"])
    C_ref = gold["C_%d" % K]
    f_ref, slack = (float(x) for x in gold["f_slack_%d" % K])
    assert abs(slack) < 2e-8           # SLSQP's own feasibility (300 unknowns at K 24)
    chol = lambda A: torch.linalg.cholesky(torch.as_tensor(A))[None].cuda()
    pl = ops.kl_cov_projection(chol(S), chol(S_old), eps)[0].cpu()
    proj = pl @ pl.T
    f = float(kl_cov(proj, torch.as_tensor(S)))
    c = float(kl_cov(proj, torch.as_tensor(S_old)))
    assert abs(c - eps) < 1e-9
    # SLSQP sits `slack` outside / inside the region, worth eta * slack of
    # objective (eta = O(10) here)
    assert abs(f - f_ref) <= 1e-8 * max(1.0, abs(f_ref)) + 20 * abs(slack)
    np.testing.assert_allclose(proj.numpy(), C_ref, rtol=5e-5, atol=5e-6)


@pytest.mark.parametrize("N,K", [(1, 1), (1, 64), (2, 2), (15, 3), (17, 64),
                                 (33, 63), (130, 24), (4096, 24)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_shared_factor_lane_vector_kernels_edge_shapes(ops, N, K, dtype):
    """vec_env_shared_kernel (wave = env, lane = vector element; csrc/lanevec.h)
    at the edges of its mapping: one env, odd env counts (a wave's second env
    missing, partial last block), K = 1 and K = 64 (every lane) -- maha,
    log-prob and the mean projection with their gradients against the
    float64 oracle formulas."""
    g = torch.Generator().manual_seed(1000 * K + N)
    L = rand_chol(K, 0.7, g, 1)[0]
    x = torch.randn(N, K, generator=g, dtype=F64)
    y = x + 0.4 * torch.randn(N, K, generator=g, dtype=F64)
    w = torch.randn(N, generator=g, dtype=F64)
    wk = torch.randn(N, K, generator=g, dtype=F64)
    Lf = L.expand(N, -1, -1)
    tol = dict(rtol=3e-4, atol=3e-4) if dtype == torch.float32 else \
        dict(rtol=1e-9, atol=1e-9)
    dev = lambda t: t.to(dtype).cuda()
    Lg = ops.expand_shared(dev(L), N)
    # maha + gradient
    xc = x.clone().requires_grad_(True)
    mc = O.maha(xc, y, Lf)
    (mc * w).sum().backward()
    xg = dev(x).requires_grad_(True)
    mg = ops.maha(xg, dev(y), Lg)
    (mg * dev(w)).sum().backward()
    torch.testing.assert_close(mg.double().cpu(), mc.detach(), **tol)
    torch.testing.assert_close(xg.grad.double().cpu(), xc.grad, **tol)
    # log-prob + gradients w.r.t. mean and the shared factor
    mean_c = y.clone().requires_grad_(True)
    L_c = L.clone().requires_grad_(True)
    lp_c = torch.distributions.MultivariateNormal(
        mean_c, scale_tril=L_c.expand(N, -1, -1)).log_prob(x)
    (lp_c * w).sum().backward()
    mean_g = dev(y).requires_grad_(True)
    L_g = dev(L).requires_grad_(True)
    lp_g = ops.mvn_log_prob(dev(x), mean_g, ops.expand_shared(L_g, N))
    (lp_g * dev(w)).sum().backward()
    torch.testing.assert_close(lp_g.double().cpu(), lp_c.detach(),
                               rtol=tol["rtol"], atol=tol["atol"] * K)
    torch.testing.assert_close(mean_g.grad.double().cpu(), mean_c.grad, **tol)
    scale = max(1.0, L_c.grad.abs().max().item())
    torch.testing.assert_close(torch.tril(L_g.grad.double().cpu()),
                               torch.tril(L_c.grad), rtol=tol["rtol"],
                               atol=tol["atol"] * scale)
    # mean projection (some rows active, some not) + gradient
    eps = 0.5 * float(mc.detach().median()) * 0.5 + 1e-6
    mu_c = x.clone().requires_grad_(True)
    maha_c, _ = KO.gaussian_kl(mu_c, Lf, y, Lf)
    pm_c = KO.mean_projection(mu_c, y, maha_c, eps)
    (pm_c * wk).sum().backward()
    mu_g = dev(x).requires_grad_(True)
    pm_g = ops.kl_mean_projection(mu_g, dev(y), Lg, eps)
    (pm_g * dev(wk)).sum().backward()
    torch.testing.assert_close(pm_g.double().cpu(), pm_c.detach(), **tol)
    gs = max(1.0, mu_c.grad.abs().max().item())
    torch.testing.assert_close(mu_g.grad.double().cpu(), mu_c.grad,
                               rtol=tol["rtol"], atol=tol["atol"] * gs)


@pytest.mark.parametrize("contextual", [False, True])
@pytest.mark.parametrize("entropy_eq", [False, True])
def test_projection_layer_entropy_first(contextual, entropy_eq):
    """KLProjectionLayer with entropy_first: the entropy control runs BEFORE the
    trust region step (no shipped config; oracle/kl_oracle.project).  Forward
    and gradients w.r.t. mean and factor, shared and per-env covariance."""
    import types
    from tce_rl_amd import ops
    from tce_rl_amd.rl.projection import KLProjectionLayer
    g = torch.Generator().manual_seed(5 + contextual)
    N, K = 12, 10
    B = N if contextual else 1
    L_o = rand_chol(K, 1.0, g, B)
    L = rand_chol(K, 0.7, g, B)             # lower entropy than the bound
    mu_o = torch.randn(N, K, generator=g, dtype=F64)
    mu = mu_o + 0.2 * torch.randn(N, K, generator=g, dtype=F64)
    beta = float(KO.entropy(L).mean()) + 0.3
    Wm = torch.randn(N, K, generator=g, dtype=F64)
    WL = torch.randn(N, K, K, generator=g, dtype=F64)
    full = lambda t: t if contextual else t.expand(N, -1, -1)
    # oracle
    mu_c, L_c = mu.clone().requires_grad_(True), L.clone().requires_grad_(True)
    pm_c, pL_c = KO.project(mu_c, full(L_c), mu_o, full(L_o), 0.05, 5e-3,
                            torch.tensor(beta, dtype=F64),
                            contextual_std=contextual, entropy_eq=entropy_eq,
                            entropy_first=True)
    ((pm_c * Wm).sum() + (full(pL_c) * WL).sum()).backward()
    # layer
    layer = KLProjectionLayer(proj_type="kl", mean_bound=0.05, cov_bound=5e-3,
                              entropy_schedule="linear", action_dim=K,
                              total_train_steps=10, target_entropy=0.0,
                              entropy_eq=entropy_eq, entropy_first=True,
                              dtype=F64, cpu=False)
    pol = types.SimpleNamespace(contextual_std=contextual)
    mu_g = mu.cuda().requires_grad_(True)
    L_g = L.cuda().requires_grad_(True)
    Lg_in = L_g if contextual else ops.expand_shared(L_g[0], N)
    Lo_in = L_o.cuda() if contextual else ops.expand_shared(L_o[0].cuda(), N)
    pm, pL = layer._projection(pol, (mu_g, Lg_in), (mu_o.cuda(), Lo_in), 0.05,
                               5e-3, torch.tensor(beta, dtype=F64).cuda())
    pL_full = ops.full_L(pL, N)
    torch.testing.assert_close(pm.cpu(), pm_c.detach(), rtol=1e-9, atol=1e-10)
    torch.testing.assert_close(pL_full.cpu(), full(pL_c).detach(), rtol=1e-8,
                               atol=1e-9)
    ((pm * Wm.cuda()).sum() + (pL_full * WL.cuda()).sum()).backward()
    torch.testing.assert_close(mu_g.grad.cpu(), mu_c.grad, rtol=1e-7, atol=1e-8)
    torch.testing.assert_close(torch.tril(L_g.grad.cpu()), torch.tril(L_c.grad),
                               rtol=1e-6, atol=1e-7)
