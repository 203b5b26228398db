"""Independent pin of the KL trust-region projection: the
oracle's closed-form / dual-root projection against a DIRECT constrained
minimiser (scipy SLSQP on the Cholesky parameters of the projected covariance)
of the problem the layer is defined by (Otto et al., ICLR 2021, eq. 5 / App. B):

    min_{S~}  KL_cov(S~ || S)        s.t.  KL_cov(S~ || S_old) <= eps_S
    min_{m~}  (m~ - m)^T S_old^-1 (m~ - m)   s.t.  maha(m~, m_old) <= eps_mu

with KL_cov(A || B) = 1/2 [tr(B^-1 A) - K + logdet B - logdet A].  Nothing of
oracle/kl_oracle.py is used to set the problem up or to solve it."""
import numpy as np
import pytest
import torch
from scipy.optimize import minimize

from oracle import kl_oracle as KO


def spd(K, g, scale=1.0):
    A = g.normal(size=(K, K))
    return scale * (A @ A.T / K + 0.5 * np.eye(K))


def kl_cov(A, B):
    """1/2 [tr(B^-1 A) - K + logdet B - logdet A] for torch float64 SPD."""
    K = A.shape[-1]
    return 0.5 * (torch.trace(torch.linalg.solve(B, A)) - K
                  + torch.logdet(B) - torch.logdet(A))


def direct_cov_projection(S, S_old, eps):
    """SLSQP over the lower-triangular factor (log-diagonal) of S~."""
    K = S.shape[0]
    St, So = torch.as_tensor(S), torch.as_tensor(S_old)
    il = np.tril_indices(K, -1)

    def build(x):
        L = torch.zeros(K, K, dtype=torch.float64)
        L[range(K), range(K)] = torch.exp(x[:K])
        L[il[0], il[1]] = x[K:]
        return L @ L.T

    def fg(fun):
        def val(x):
            xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
            return float(fun(build(xt)).detach())

        def grad(x):
            xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
            fun(build(xt)).backward()
            return xt.grad.numpy()
        return val, grad
    obj, obj_g = fg(lambda C: kl_cov(C, St))
    con, con_g = fg(lambda C: eps - kl_cov(C, So))
    Lo = np.linalg.cholesky(S_old)                       # feasible start
    x0 = np.concatenate([np.log(np.diag(Lo)), Lo[il]])
    best = None
    for _ in range(3):                                   # restarts tighten it
        r = minimize(obj, x0, jac=obj_g, method="SLSQP",
                     constraints=[{"type": "ineq", "fun": con, "jac": con_g}],
                     options={"ftol": 1e-16, "maxiter": 2000})
        x0 = r.x
        best = r
    C = build(torch.tensor(best.x, dtype=torch.float64)).numpy()
    return C, best.fun, con(best.x)


@pytest.mark.parametrize("K", [4, 12])
@pytest.mark.parametrize("seed", [0, 1])
def test_cov_projection_is_the_constrained_optimum(K, seed):
    g = np.random.default_rng(seed)
    S_old = spd(K, g)
    S = spd(K, g, scale=1.7)                 # far from S_old: constraint active
    eps = 5e-3
    C_ref, f_ref, slack = direct_cov_projection(S, S_old, eps)
    assert abs(slack) < 1e-9                 # active at the optimum
    cov = torch.as_tensor(S)[None]
    L_old = torch.linalg.cholesky(torch.as_tensor(S_old))[None]
    proj, eta = KO.cov_projection(cov, L_old, eps)
    proj = proj[0]
    assert eta.item() > 0
    f = float(kl_cov(proj, torch.as_tensor(S)))
    c = float(kl_cov(proj, torch.as_tensor(S_old)))
    assert abs(c - eps) < 1e-9               # tight
    assert abs(f - f_ref) <= 1e-8 * max(1.0, abs(f_ref))   # same optimum value
    # (SLSQP may sit up to its 1e-9 constraint slack outside the region, which
    # buys it eta * slack of objective)
    assert f <= f_ref + 1e-8 * max(1.0, abs(f_ref))
    # a strictly convex problem: same minimiser (SLSQP's argmin is accurate to
    # ~sqrt(its objective accuracy))
    np.testing.assert_allclose(proj.numpy(), C_ref, rtol=2e-5, atol=2e-6)


def test_cov_projection_inactive_is_identity_for_the_direct_problem():
    g = np.random.default_rng(3)
    K = 6
    S_old = spd(K, g)
    S = S_old * 1.01
    eps = 5e-3
    assert float(kl_cov(torch.as_tensor(S), torch.as_tensor(S_old))) < eps
    C_ref, f_ref, slack = direct_cov_projection(S, S_old, eps)
    assert slack > 0 and f_ref < 1e-12       # unconstrained optimum S~ = S
    proj, eta = KO.cov_projection(torch.as_tensor(S)[None], torch.linalg.cholesky(
        torch.as_tensor(S_old))[None], eps)
    assert eta.item() == 0
    np.testing.assert_allclose(proj[0].numpy(), S, rtol=0, atol=0)
    np.testing.assert_allclose(C_ref, S, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("K", [4, 12])
def test_mean_projection_is_the_constrained_optimum(K):
    g = np.random.default_rng(7)
    S_old = spd(K, g)
    P = np.linalg.inv(S_old)
    m_old = g.normal(size=K)
    m = m_old + g.normal(size=K)
    eps = 0.01                                # maha bound as the layer uses it:
    # gaussian_kl's mean part is 1/2 maha, the bound applies to that part

    def half_maha(a, b):
        d = a - b
        return 0.5 * d @ P @ d
    r = minimize(lambda x: half_maha(x, m), m_old,
                 jac=lambda x: P @ (x - m), method="SLSQP",
                 constraints=[{"type": "ineq",
                               "fun": lambda x: eps - half_maha(x, m_old),
                               "jac": lambda x: -P @ (x - m_old)}],
                 options={"ftol": 1e-16, "maxiter": 1000})
    L_old = torch.linalg.cholesky(torch.as_tensor(S_old))[None]
    mt, mot = torch.as_tensor(m)[None], torch.as_tensor(m_old)[None]
    maha_part, _ = KO.gaussian_kl(mt, L_old, mot, L_old)
    pm = KO.mean_projection(mt, mot, maha_part, eps)[0].numpy()
    assert abs(half_maha(pm, m_old) - eps) < 1e-12
    assert abs(half_maha(pm, m) - r.fun) <= 1e-8 * max(1.0, r.fun)
    np.testing.assert_allclose(pm, r.x, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("K", [4, 12])
def test_slsqp_fixture_is_reproducible(K):
    """tests/golden/kl_slsqp.npz (read by the GPU test of the projection
    kernel) holds what direct_cov_projection returns for the same inputs."""
    import os
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "kl_slsqp.npz"))
    g = np.random.default_rng(K)
    S_old, S = spd(K, g), spd(K, g, scale=1.7)
    np.testing.assert_array_equal(S, gold["S_%d" % K])
    np.testing.assert_array_equal(S_old, gold["S_old_%d" % K])
    C, f, slack = direct_cov_projection(S, S_old, float(gold["eps"]))
    np.testing.assert_allclose(C, gold["C_%d" % K], rtol=1e-6, atol=1e-8)
    assert abs(f - float(gold["f_slack_%d" % K][0])) <= 1e-8 * max(1.0, abs(f))
