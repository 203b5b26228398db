"""CPU restatement of the Frobenius trust-region projection (Otto et al.,
"Differentiable Trust Region Layers for Deep Reinforcement Learning", ICLR
2021, section 4.1 and appendix B.1).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

parity unpinned: the class the reference's factory names
(``mprl/rl/projection/__init__.py:4-5,18-24``, ``FrobeniusProjectionLayer``)
lives in the un-vendored dependency ``BruceGeLi/trust-region-layers`` @
``TCE_ICLR24`` (``conda_env.sh:56-60``) and no experiment file of the reference
selects it, so neither its source nor a vector of it exists here; this file
restates the paper, sample by sample, in float64:

  d_mean = (mu_o - mu)^T Sigma_o^-1 (mu_o - mu)     (scale_prec; else |mu_o - mu|^2)
  d_cov  = tr((Sigma_o - Sigma)^T (Sigma_o - Sigma))
  mu~    = (mu + omega mu_o) / (1 + omega),    omega = sqrt(d_mean / eps_mu) - 1   if d_mean > eps_mu
  Sigma~ = (Sigma + eta Sigma_o) / (1 + eta),  eta   = sqrt(d_cov / eps_S) - 1      if d_cov  > eps_S
and returns the Cholesky factor of Sigma~.  Self-checks standing in for goldens
(tests/test_frob_gpu.py): after the projection both distances equal their
bounds where they were exceeded (the closed forms scale the differences by 1 /
(1 + omega) resp. 1 / (1 + eta)), and the layer is the identity inside them.
"""
import torch


def metric(mean, L, mean_o, L_o, scale_prec=True):
    """(d_mean, d_cov) for ONE sample (1-D mean, 2-D factors)."""
    d = mean_o - mean
    if scale_prec:
        z = torch.linalg.solve_triangular(L_o, d[:, None], upper=False)
        d_mean = (z * z).sum()
    else:
        d_mean = (d * d).sum()
    diff = L_o @ L_o.T - L @ L.T
    return d_mean, (diff * diff).sum()


def project(mean, L, mean_o, L_o, eps, eps_cov, scale_prec=True,
            contextual_std=True):
    """mean [N, K], L [N, K, K] (all rows equal when not contextual_std: the
    first matrix is projected and broadcast) -> (proj_mean, proj_L)."""
    N = mean.shape[0]
    means, Ls = [], []
    for n in range(N):
        k = n if contextual_std else 0
        d_mean, d_cov = metric(mean[n], L[k], mean_o[n], L_o[k], scale_prec)
        if d_mean > eps:
            omega = torch.sqrt(d_mean / eps) - 1.0
            means.append((mean[n] + omega * mean_o[n]) / (1.0 + omega))
        else:
            means.append(mean[n])
        if d_cov > eps_cov:
            eta = torch.sqrt(d_cov / eps_cov) - 1.0
            cov = (L[k] @ L[k].T + eta * (L_o[k] @ L_o[k].T)) / (1.0 + eta)
            Ls.append(torch.linalg.cholesky(cov))
        else:
            Ls.append(L[k])
    return torch.stack(means), torch.stack(Ls)
