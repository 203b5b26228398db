"""CPU restatement of the KL trust-region projection layer.  No reference vectors
exist (third-party arithmetic, absent): pinned against an independent SLSQP
solution of the constrained problem instead (tests/test_kl_optimum_cpu.py).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

The arithmetic is not in ``/root/reference``: it lives in the un-vendored
dependencies ``BruceGeLi/trust-region-layers`` @ ``TCE_ICLR24``
(``conda_env.sh:56-60``, ``README.md:120-123``) and its C++ dual solver
``cppprojection`` / ITPAL (``conda_env.sh:34``, ``README.md:54-69``).  This file
restates the published algorithm (Otto et al., "Differentiable Trust Region
Layers for Deep Reinforcement Learning", ICLR 2021) behind the reference's call
sites:

* factory / kwargs            ``mprl/rl/projection/__init__.py:18-40``,
                              ``mprl/config/metaworld/tcp/entire/shared.yaml:105-120``
* ``projection(policy, p, q, step)``, ``initial_entropy``,
  ``get_trust_region_loss``   ``mprl/rl/agent/temporal_correlated_agent.py:439-441,530-567``
* ``gaussian_kl_details``     ``mprl/rl/agent/temporal_correlated_agent.py:641-686``
* ``compute_metrics``         ``mprl/rl/agent/black_box_agent.py:359-363``
* policy call-backs used      ``mprl/rl/policy/black_box_policy.py:156-224``

Algorithm
  mean:  m = (mu-mu_o)^T Sigma_o^-1 (mu-mu_o); if m > eps_mu:
         omega = sqrt(m/eps_mu) - 1, mu~ = (mu + omega mu_o)/(1 + omega)
  cov:   c(Sigma) = 1/2 [tr(Sigma_o^-1 Sigma) - K + logdet Sigma_o - logdet Sigma];
         if c > eps_S: Sigma~^-1 = (eta Sigma_o^-1 + Sigma^-1)/(eta + 1) with
         eta > 0 the root of c(Sigma~(eta)) = eps_S (the convex dual's
         stationarity condition); gradient by implicit differentiation.
         Non-contextual covariance: only the first matrix is projected and the
         result is broadcast over the batch.
  entropy control: after the trust region step, if H < beta(step) scale L by
         exp((beta - H)/K)  (inequality form; equality form scales always).
  trust-region loss: coeff * mean(d_mean(p, proj.detach) [+ d_cov]).
Self-checks standing in for goldens: tests/test_kl_oracle.py (tightness of the
bounds, identity when inactive, gradient vs finite differences).
"""
import math

import torch


def gaussian_kl(mean, L, mean_o, L_o):
    """(maha_part, cov_part) of KL(N(mean, LL^T) || N(mean_o, L_o L_o^T))."""
    K = mean.shape[-1]
    logdet = 2 * L.diagonal(dim1=-2, dim2=-1).log().sum(-1)
    logdet_o = 2 * L_o.diagonal(dim1=-2, dim2=-1).log().sum(-1)
    diff = (mean - mean_o)[..., None]
    maha = torch.linalg.solve_triangular(L_o, diff, upper=False) \
        .pow(2).sum([-2, -1])
    # tr(Sigma_o^-1 Sigma) = || L_o^-1 L ||_F^2
    A = torch.linalg.solve_triangular(L_o, L, upper=False)
    trace = A.pow(2).sum([-2, -1])
    return 0.5 * maha, 0.5 * (trace - K + logdet_o - logdet)


def gaussian_kl_details(mean, L, mean_o, L_o):
    """mean / cov / shape / volume parts (cov = shape + volume)."""
    K = mean.shape[-1]
    logdet = 2 * L.diagonal(dim1=-2, dim2=-1).log().sum(-1)
    logdet_o = 2 * L_o.diagonal(dim1=-2, dim2=-1).log().sum(-1)
    maha_part, cov_part = gaussian_kl(mean, L, mean_o, L_o)
    volume = 0.5 * (logdet_o - logdet)
    shape = cov_part - volume
    return maha_part, cov_part, shape, volume


def entropy(L):
    K = L.shape[-1]
    return 0.5 * K * (1 + math.log(2 * math.pi)) + \
        L.diagonal(dim1=-2, dim2=-1).log().sum(-1)


def entropy_schedule(kind, initial_entropy, target_entropy, temperature, step,
                     total_train_steps, dim):
    if kind == "linear":
        return step * (target_entropy - initial_entropy) / total_train_steps \
            + initial_entropy
    if kind == "exp":
        return dim * target_entropy + (initial_entropy - dim * target_entropy) \
            * temperature ** (10 * step / total_train_steps)
    return torch.as_tensor(-float("inf"))


def mean_projection(mean, mean_o, maha, eps):
    mask = maha > eps
    omega = torch.ones_like(maha)
    # masked assignment (not torch.where): sqrt'(0) = inf must not reach the
    # rows that are not projected
    omega[mask] = torch.sqrt(maha[mask] / eps) - 1.0
    omega = torch.max(-omega, omega)[..., None]
    m = (mean + omega * mean_o) / (1 + omega + 1e-16)
    return torch.where(mask[..., None], m, mean)


def _kl_of_eta(eta, lam):
    mu = (eta + 1) * lam / (eta * lam + 1)
    return 0.5 * (mu - 1 - mu.log()).sum(-1)


def solve_eta(cov, L_o, eps):
    """eta >= 0 per matrix (0 where the constraint is inactive); float64."""
    c64, Lo64 = cov.double(), L_o.double()
    W = torch.linalg.solve_triangular(Lo64, c64, upper=False)
    B = torch.linalg.solve_triangular(Lo64, W.transpose(-1, -2), upper=False)
    lam = torch.linalg.eigvalsh(0.5 * (B + B.transpose(-1, -2)))
    eta = torch.zeros(cov.shape[:-2], dtype=torch.float64)
    for i in range(lam.shape[0]) if lam.ndim > 1 else [None]:
        l = lam if i is None else lam[i]
        if _kl_of_eta(torch.tensor(0.0, dtype=torch.float64), l) <= eps:
            continue
        lo, hi = 0.0, 1.0
        while _kl_of_eta(torch.tensor(hi, dtype=torch.float64), l) > eps:
            lo, hi = hi, hi * 2
        for _ in range(200):
            mid = 0.5 * (lo + hi)
            if _kl_of_eta(torch.tensor(mid, dtype=torch.float64), l) > eps:
                lo = mid
            else:
                hi = mid
        if i is None:
            eta = torch.tensor(0.5 * (lo + hi), dtype=torch.float64)
        else:
            eta[i] = 0.5 * (lo + hi)
    return eta


def cov_projection(cov, L_o, eps):
    """Projected covariance, differentiable w.r.t. ``cov`` (implicit gradient
    through eta via one differentiable Newton step at the converged root)."""
    K = cov.shape[-1]
    eta0 = solve_eta(cov.detach(), L_o.detach(), eps).to(cov.dtype)
    active = eta0 > 0
    prec_o = torch.cholesky_inverse(L_o)
    logdet_o = 2 * L_o.diagonal(dim1=-2, dim2=-1).log().sum(-1)
    prec_t = torch.linalg.inv(cov)

    def kl_at(eta, prec_t_):
        e = eta[..., None, None]
        prec = (e * prec_o + prec_t_) / (e + 1)
        cov_p = torch.linalg.inv(prec)
        tr = (prec_o * cov_p.transpose(-1, -2)).sum([-2, -1])
        return 0.5 * (tr - K + logdet_o + torch.logdet(prec)), cov_p

    eta_var = eta0.clone().requires_grad_(True)
    with torch.enable_grad():
        kl_d, _ = kl_at(eta_var, prec_t.detach())
        dkl_deta, = torch.autograd.grad(kl_d.sum(), eta_var)
    kl_v, _ = kl_at(eta0, prec_t)
    safe = torch.where(active, dkl_deta, torch.ones_like(dkl_deta))
    eta = eta0 - torch.where(active, (kl_v - eps) / safe,
                             torch.zeros_like(kl_v))
    _, cov_p = kl_at(eta, prec_t)
    return torch.where(active[..., None, None], cov_p, cov), eta0


def kl_projection(mean, L, mean_o, L_o, mean_bound, cov_bound,
                  contextual_std=True):
    """Trust-region part: (proj_mean, proj_L)."""
    if not contextual_std:
        Lp, L_op = L[:1], L_o[:1]
    else:
        Lp, L_op = L, L_o
    maha_part, _ = gaussian_kl(mean, L, mean_o, L_o)
    proj_mean = mean_projection(mean, mean_o, maha_part, mean_bound)
    cov = torch.einsum('...ij,...kj->...ik', Lp, Lp)
    proj_cov, _ = cov_projection(cov, L_op, cov_bound)
    proj_L = torch.linalg.cholesky(proj_cov)
    if not contextual_std:
        proj_L = proj_L.expand(mean.shape[0], -1, -1)
    return proj_mean, proj_L


def entropy_projection(mean, L, beta, equality=False):
    K = L.shape[-1]
    ent = entropy(L)
    beta = torch.as_tensor(beta, dtype=L.dtype).expand_as(ent)
    alpha = torch.exp((beta - ent) / K)
    if not equality:
        alpha = torch.where(ent < beta, alpha, torch.ones_like(alpha))
    return mean, L * alpha[..., None, None]


def project(mean, L, mean_o, L_o, mean_bound, cov_bound, beta,
            contextual_std=True, entropy_eq=False, entropy_first=False):
    """Full layer call: trust region + entropy control."""
    if entropy_first:
        mean, L = entropy_projection(mean, L, beta, entropy_eq)
    pm, pL = kl_projection(mean, L, mean_o, L_o, mean_bound, cov_bound,
                           contextual_std)
    if entropy_first:
        return pm, pL
    return entropy_projection(pm, pL, beta, entropy_eq)


def trust_region_loss(mean, L, proj_mean, proj_L, coeff, include_cov):
    md, cd = gaussian_kl(mean, L, proj_mean.detach(), proj_L.detach())
    return (md + cd if include_cov else md).mean() * coeff
