"""KL trust-region projection layer (differentiable), MI355X build.

The reference imports this layer from the third-party package
``trust_region_projections`` (+ the C++ dual solver ``cpp_projection``/ITPAL);
see ``mprl/rl/projection/__init__.py:18-40`` for the factory / kwargs and
``mprl/rl/agent/temporal_correlated_agent.py:439-441,530-567,641-686`` for the
call sites this class serves:

    proj_mean, proj_L = projection(policy, (mean, L), (old_mean, old_L), step)
    projection.initial_entropy                      (latched once)
    projection.get_trust_region_loss(policy, p, proj_p, set_variance=...)
    projection.compute_metrics(policy, p, q, step)  (BBRL)
    gaussian_kl_details(policy, p, q) -> mean, cov, shape, volume parts

Algorithm (Otto et al., ICLR 2021; restated in oracle/kl_oracle.py): closed
form mean projection on the Mahalanobis part, covariance projection by
interpolating precisions with the multiplier eta that makes the KL constraint
tight, gradient by implicit differentiation; entropy control afterwards.
The covariance step runs in ``ops.kl_cov_projection`` (one workgroup per
matrix; with a non-contextual covariance only ONE matrix is projected and the
result is broadcast, exactly like the reference layer does).
"""
import numpy as np
import torch

from .. import ops
from ..util import parse_dtype_device


def gaussian_kl(policy, p, q):
    """(maha_part, cov_part) [N] of KL(p || q)."""
    mean, L = p
    mean_o, L_o = q
    maha_part = 0.5 * policy.maha(mean, mean_o, L_o)
    cov_part = ops.kl_cov_part(L, L_o, mean.shape[0])
    return maha_part, cov_part


def gaussian_kl_details(policy, p, q):
    """mean / cov / shape / volume parts; cov = shape + volume."""
    mean, L = p
    mean_o, L_o = q
    maha_part, cov_part = gaussian_kl(policy, p, q)
    volume = 0.5 * (policy.log_determinant(L_o) - policy.log_determinant(L))
    return maha_part, cov_part, cov_part - volume, volume


def get_entropy_schedule(kind, total_train_steps, dim):
    if kind == "linear":
        return lambda init, target, temp, step: \
            step * (target - init) / total_train_steps + init
    if kind == "exp":
        return lambda init, target, temp, step: dim * target + \
            (init - dim * target) * temp ** (10 * step / total_train_steps)
    return lambda init, target, temp, step: init.new_full((), -np.inf)


class BaseProjectionLayer:
    def __init__(self, proj_type="", mean_bound=0.03, cov_bound=1e-3,
                 trust_region_coeff=0.0, scale_prec=True,
                 entropy_schedule=None, action_dim=None,
                 total_train_steps=None, target_entropy=0.0, temperature=0.5,
                 entropy_eq=False, entropy_first=False, do_regression=False,
                 cpu=True, dtype=torch.float32, **unused):
        self.proj_type = proj_type
        self.mean_bound = float(mean_bound)
        self.cov_bound = float(cov_bound)
        self.trust_region_coeff = float(trust_region_coeff)
        self.scale_prec = scale_prec
        assert (action_dim and total_train_steps) if entropy_schedule else True
        self.entropy_schedule_type = entropy_schedule
        self.entropy_schedule = get_entropy_schedule(
            entropy_schedule, total_train_steps, action_dim)
        self.target_entropy = float(target_entropy)
        self.temperature = float(temperature)
        self.entropy_eq, self.entropy_first = entropy_eq, entropy_first
        # accepted and stored like the third-party constructor does; the flag
        # only matters to that layer's `trust_region_regression` step, which the
        # reference's agents never call (no call site under mprl/)
        self.do_regression = do_regression
        self.dtype = dtype
        self._initial_entropy = None

    @property
    def initial_entropy(self):
        return self._initial_entropy

    @initial_entropy.setter
    def initial_entropy(self, entropy):
        if self._initial_entropy is None:
            self._initial_entropy = entropy

    def __call__(self, policy, p, q, step, *args, **kwargs):
        init = self.initial_entropy
        if init is None:
            init = p[0].new_full((), -np.inf)
        beta = self.entropy_schedule(init, self.target_entropy,
                                     self.temperature, step)
        return self._projection(policy, p, q, self.mean_bound, self.cov_bound,
                                beta)

    def _trust_region_projection(self, policy, p, q, eps, eps_cov, beta):
        return p

    def _projection(self, policy, p, q, eps, eps_cov, beta):
        """Trust region step, then entropy control (entropy_first=False, the
        setting of every shipped config; the scaling is fused into the
        covariance kernel) -- or the other way round."""
        if self.entropy_schedule_type in (None, False):
            beta = None                                # bound is -inf
        if self.entropy_first and beta is not None:
            # entropy control BEFORE the trust region step (no shipped config;
            # a few elementwise torch ops on the factor, autograd through them)
            p = self._entropy_projection(policy, p, beta)
            beta = None
        return self._trust_region_projection(policy, p, q, eps, eps_cov, beta)

    def _entropy_projection(self, policy, p, beta):
        """Scale the factor by alpha = exp((beta - H) / K) where the entropy H
        is below the bound beta (entropy_eq: everywhere)."""
        mean, L = p
        N, K = mean.shape
        shared = not policy.contextual_std
        Lb = ops.first_matrix(L) if shared else ops.full_L(L, N)
        ent = 0.5 * K * (1.0 + np.log(2.0 * np.pi)) + \
            Lb.diagonal(dim1=-2, dim2=-1).log().sum(-1)
        b = beta.detach().to(ent.dtype).reshape(()).expand_as(ent)
        alpha = torch.exp((b - ent) / K)
        if not self.entropy_eq:
            alpha = torch.where(ent < b, alpha, torch.ones_like(alpha))
        Ls = Lb * alpha[..., None, None]
        return mean, (ops.expand_shared(Ls, N) if shared else Ls)

    def trust_region_value(self, policy, p, q):
        return gaussian_kl(policy, p, q)

    def get_trust_region_loss(self, policy, p, proj_p, set_variance=False):
        target = (proj_p[0].detach(), ops.detach_L(proj_p[1]))
        mean_diff, cov_diff = self.trust_region_value(policy, p, target)
        # a directly-set (non-contextual, set_variance) covariance needs no
        # regression term
        if policy.contextual_std or not set_variance:
            delta = (mean_diff + cov_diff).mean()
        else:
            delta = mean_diff.mean()
        return delta * self.trust_region_coeff

    @torch.no_grad()
    def compute_metrics(self, policy, p, q, step=None):
        entropy = policy.entropy(p)
        mean_kl, cov_kl = gaussian_kl(policy, p, q)
        kl = mean_kl + cov_kl
        init = self.initial_entropy if self.initial_entropy is not None \
            else entropy.mean()
        return {"kl": kl.mean(), "constraint": kl.mean(),
                "mean_constraint": mean_kl.mean(),
                "cov_constraint": cov_kl.mean(), "entropy": entropy.mean(),
                "entropy_diff": (init - entropy).mean(), "kl_max": kl.max(),
                "constraint_max": kl.max(),
                "mean_constraint_max": mean_kl.max(),
                "cov_constraint_max": cov_kl.max(),
                "entropy_max": entropy.max()}


class KLProjectionLayer(BaseProjectionLayer):
    def _trust_region_projection(self, policy, p, q, eps, eps_cov, beta):
        mean, L = p
        mean_o, L_o = q
        N = mean.shape[0]
        proj_mean = ops.kl_mean_projection(mean, mean_o, L_o, eps)
        if not policy.contextual_std:
            # only ONE matrix is projected, then broadcast over the batch
            Lb, Lob = ops.first_matrix(L), ops.first_matrix(L_o)
            proj_base = ops.kl_cov_projection(Lb[None], Lob[None], eps_cov,
                                              beta, self.entropy_eq)[0]
            proj_L = ops.expand_shared(proj_base, N)
        else:
            proj_L = ops.kl_cov_projection(ops.full_L(L, N), ops.full_L(L_o, N),
                                           eps_cov, beta, self.entropy_eq)
        return proj_mean, proj_L


def gaussian_frobenius(policy, p, q, scale_prec):
    """(mean_part, cov_part) [N] of the Frobenius trust-region metric of Otto
    et al. (ICLR 2021, sec. 4.1): mean_part = (mu_q - mu)^T Sigma_q^-1 (mu_q -
    mu) (scale_prec; else the squared Euclidean distance), cov_part =
    tr((Sigma_q - Sigma)^T (Sigma_q - Sigma))."""
    mean, L = p
    mean_o, L_o = q
    N = mean.shape[0]
    if scale_prec:
        mean_part = ops.maha(mean, mean_o, L_o)
    else:
        mean_part = ((mean_o - mean) ** 2).sum(-1)
    shared = getattr(L, "_tce_base", None) is not None or L.dim() == 2
    Lb = ops.first_matrix(L)[None] if shared else ops.full_L(L, N)
    Lob = ops.first_matrix(L_o)[None] if (
        getattr(L_o, "_tce_base", None) is not None or L_o.dim() == 2) \
        else ops.full_L(L_o, N)
    diff = Lob @ Lob.transpose(-1, -2) - Lb @ Lb.transpose(-1, -2)
    cov_part = (diff * diff).sum((-1, -2))
    return mean_part, cov_part.expand(N) if cov_part.shape[0] == 1 else cov_part


class FrobeniusProjectionLayer(BaseProjectionLayer):
    """Frobenius projection (Otto et al. 2021, sec. 4.1 / app. B.1), named by
    the reference's factory (mprl/rl/projection/__init__.py:4-5,18-24) and used
    by none of its experiment files.  Both steps are closed forms --
        mu~    = (mu + omega mu_old) / (1 + omega),   omega = sqrt(d_mean / eps_mu) - 1
        Sigma~ = (Sigma + eta Sigma_old) / (1 + eta), eta   = sqrt(d_cov / eps_Sigma) - 1
    where a bound is exceeded, the identity elsewhere -- and run as a handful of
    batched torch operations on the device with autograd through them (one
    matrix when the covariance is not contextual); not a hot path of any
    BASELINE config, so no kernel was written for it.  The third-party class is
    not under /root/reference: what is built is the paper's statement."""

    def trust_region_value(self, policy, p, q):
        return gaussian_frobenius(policy, p, q, self.scale_prec)

    def _trust_region_projection(self, policy, p, q, eps, eps_cov, beta):
        mean, L = p
        mean_o, L_o = q
        N = mean.shape[0]
        mean_o, L_o = mean_o.detach(), ops.detach_L(L_o)
        mean_part, cov_part = gaussian_frobenius(policy, (mean, L), (mean_o, L_o),
                                                 self.scale_prec)
        omega = torch.sqrt(mean_part.clamp_min(1e-30) / eps) - 1.0
        omega = torch.where(mean_part > eps, omega, torch.zeros_like(omega))
        proj_mean = (mean + omega[:, None] * mean_o) / (1.0 + omega[:, None])
        shared = not policy.contextual_std
        Lb = ops.first_matrix(L)[None] if shared else ops.full_L(L, N)
        Lob = ops.first_matrix(L_o)[None] if shared else ops.full_L(L_o, N)
        cp = cov_part[:1] if shared else cov_part
        cov, cov_o = Lb @ Lb.transpose(-1, -2), Lob @ Lob.transpose(-1, -2)
        eta = torch.sqrt(cp.clamp_min(1e-30) / eps_cov) - 1.0
        eta = torch.where(cp > eps_cov, eta, torch.zeros_like(eta))
        new_cov = (cov + eta[:, None, None] * cov_o) / (1.0 + eta[:, None, None])
        proj = torch.where((cp > eps_cov)[:, None, None],
                           torch.linalg.cholesky(new_cov), Lb)
        proj_p = (proj_mean, ops.expand_shared(proj[0], N) if shared else proj)
        if beta is not None:
            proj_p = self._entropy_projection(policy, proj_p, beta)
        return proj_p


def projection_factory(typ, **kwargs):
    """mprl/rl/projection/__init__.py:18-40."""
    dtype, device = parse_dtype_device(kwargs["dtype"], kwargs["device"])
    kwargs = dict(kwargs)
    kwargs["cpu"] = device == torch.device("cpu")
    kwargs["dtype"] = dtype
    del kwargs["device"]
    classes = {"BaseProjectionLayer": BaseProjectionLayer,
               "KLProjectionLayer": KLProjectionLayer,
               "FrobeniusProjectionLayer": FrobeniusProjectionLayer}
    if typ not in classes:
        raise NotImplementedError(
            "%s: the KL projection of every TCE / BBRL config, the Frobenius "
            "and the identity layer are built; the Wasserstein and PAPI layers "
            "of the third-party package are not" % typ)
    return classes[typ](**kwargs)
