"""Self-checks of the (unpinned) KL-projection oracle: the properties that stand
in for goldens of the third-party trust_region_projections / cpp_projection."""
import torch

from oracle import kl_oracle as KO
from oracle import tce_oracle as O


def rand_chol(K, scale, g, B=1):
    vec = torch.cat([scale * torch.randn(B, K, generator=g, dtype=torch.float64),
                     0.1 * scale * torch.randn(B, K * (K - 1) // 2, generator=g,
                                               dtype=torch.float64)], -1)
    return O.vector_to_cholesky(vec, K, 1e-3, False)


def test_cov_projection_tight_and_identity():
    g = torch.Generator().manual_seed(0)
    K = 12
    L_o = rand_chol(K, 1.0, g)
    L_far = rand_chol(K, 1.0, g)             # far away -> active
    L_near = L_o * 1.0001                    # within the bound -> identity
    eps = 1e-3
    for L, active in ((L_far, True), (L_near, False)):
        cov = KO.covariance_like(L) if hasattr(KO, "covariance_like") else \
            L @ L.transpose(-1, -2)
        proj_cov, eta = KO.cov_projection(cov, L_o, eps)
        pl = torch.linalg.cholesky(proj_cov)
        _, kl = KO.gaussian_kl(torch.zeros(1, K, dtype=torch.float64), pl,
                               torch.zeros(1, K, dtype=torch.float64), L_o)
        if active:
            assert eta.item() > 0
            assert abs(kl.item() - eps) < 1e-9
        else:
            assert eta.item() == 0
            assert torch.allclose(proj_cov, cov)


def test_cov_projection_gradient_matches_finite_differences():
    g = torch.Generator().manual_seed(1)
    K = 6
    L_o = rand_chol(K, 1.0, g)
    L = rand_chol(K, 1.0, g).requires_grad_(True)
    W = torch.randn(K, K, generator=g, dtype=torch.float64)

    def f(Lx):
        cov = Lx @ Lx.transpose(-1, -2)
        pc, _ = KO.cov_projection(cov, L_o, 1e-2)
        return (torch.linalg.cholesky(pc) * W).sum()

    f(L).backward()
    num = torch.zeros_like(L)
    h = 1e-6
    with torch.no_grad():
        for i in range(K):
            for j in range(i + 1):
                d = torch.zeros_like(L)
                d[0, i, j] = h
                num[0, i, j] = (f(L + d) - f(L - d)) / (2 * h)
    assert torch.allclose(torch.tril(L.grad), num, rtol=1e-5, atol=1e-7)


def test_mean_projection_tight():
    g = torch.Generator().manual_seed(2)
    K, N = 8, 5
    L_o = rand_chol(K, 1.0, g, N)
    mu_o = torch.randn(N, K, generator=g, dtype=torch.float64)
    mu = mu_o + torch.randn(N, K, generator=g, dtype=torch.float64)
    maha, _ = KO.gaussian_kl(mu, L_o, mu_o, L_o)
    pm = KO.mean_projection(mu, mu_o, maha, 0.01)
    m2, _ = KO.gaussian_kl(pm, L_o, mu_o, L_o)
    assert torch.allclose(m2, torch.full_like(m2, 0.01), rtol=1e-9)


def test_entropy_projection_and_schedule():
    g = torch.Generator().manual_seed(3)
    L = rand_chol(5, 0.3, g, 4)
    beta = KO.entropy(L) + torch.tensor([0.5, -0.5, 0.1, -0.1])
    _, Lp = KO.entropy_projection(torch.zeros(4, 5), L, beta)
    ent = KO.entropy(Lp)
    assert torch.allclose(ent, torch.maximum(KO.entropy(L), beta))
    init, tgt = torch.tensor(10.0), 0.0
    assert KO.entropy_schedule("linear", init, tgt, 0.7, 0, 100, 5) == 10.0
    assert KO.entropy_schedule("linear", init, tgt, 0.7, 100, 100, 5) == 0.0
    assert KO.entropy_schedule("exp", init, tgt, 0.5, 10, 100, 5) == 5.0
