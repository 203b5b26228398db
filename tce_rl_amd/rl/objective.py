"""Fused policy objective of one TCE epoch for a shared (non-contextual)
covariance: projection -> pair log-prob -> surrogate -> trust-region loss ->
entropy / KL diagnostics as ONE autograd node made of ~12 HIP kernels.

It evaluates exactly what ``TemporalCorrelatedAgent.update_policy`` builds
from separate pieces per epoch (mprl/rl/agent/temporal_correlated_agent.py:
523-612): ``projection(policy, new, old)``, ``policy.log_prob``,
``surrogate_loss``, ``entropy_loss``, ``get_trust_region_loss``,
``kl_old_new_proj`` -- and returns their sum with the gradient w.r.t. the new
mean [N, K] and the new Cholesky factor [K, K].  The chain rule through the
projection is applied inside (the backward kernels run in ``forward``; the
upstream gradient of a scalar loss only scales the stored result), so autograd
sees one node instead of ~120 small ones.
"""
import torch

from .. import _lib, ops
from .._lib import call, ptr, sfx, stream
from .projection import KLProjectionLayer


def supported(agent, dataset):
    pol, proj = agent.policy, agent.projection
    L_old = dataset["segment_params_L"]
    return (not pol.contextual_std and type(proj) is KLProjectionLayer
            and not proj.entropy_first and not proj.do_regression
            and ops.split_L(L_old)[1] == 0
            and not (pol.mp.disable_goal or pol.mp.disable_weights))


class Context:
    """Per-update constants of the objective (dataset tensors, bounds)."""

    def __init__(self, agent, dataset, times, beta):
        c = lambda t: t if t.is_contiguous() else t.contiguous()
        pol, proj = agent.policy, agent.projection
        self.mp = pol.mp
        self.mean_old = c(dataset["segment_params_mean"])
        self.L_old = c(ops.split_L(dataset["segment_params_L"])[0].detach())
        self.traj = c(dataset["step_actions"])
        self.lp_old = c(dataset["segment_log_prob_estimate"])
        self.adv = c(dataset["segment_advantage"])
        self.t0 = c(dataset["segment_init_time"])
        self.y0 = c(dataset["segment_init_pos"])
        self.v0 = c(dataset["segment_init_vel"])
        self.times = c(times)
        self.general = 0 if getattr(times, "_tce_affine", False) else 1
        self.pairs = c(agent.sampler.pred_pairs.to(torch.int64))
        self.eps_mean, self.eps_cov = proj.mean_bound, proj.cov_bound
        self.beta = None if beta is None else \
            c(beta.detach().to(self.mean_old.dtype).reshape(1))
        self.entropy_eq = int(bool(proj.entropy_eq))
        self.tr_coeff = proj.trust_region_coeff
        self.tr_include_cov = int(pol.contextual_std or not agent.set_variance)
        self.ent_coef = float(agent.entropy_penalty_coef)
        K = self.mean_old.shape[-1]
        self.proj_ctx = torch.zeros(_lib.load().tce_kl_cov_proj_ctx_len(K),
                                    dtype=torch.float64,
                                    device=self.mean_old.device)
        # surrogate kernel scratch: block ticket (zeroed once) + partials
        self.sur_ws = torch.zeros(_lib.load().tce_surrogate_ws_len(),
                                  dtype=torch.float64,
                                  device=self.mean_old.device)


class _Objective(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mean_new, L_new, c):
        N, K = mean_new.shape
        dt, dev = mean_new.dtype, mean_new.device
        s = sfx(dt)
        st = stream()
        mp = c.mp
        T, P = c.times.shape[1], c.pairs.shape[0]
        new = lambda *shape: torch.empty(*shape, dtype=dt, device=dev)
        # ---- projection: mean (closed form per env), covariance (one matrix)
        pm = new(N, K)
        call("tce_vec_env_" + s, 1, 0, ptr(mean_new), ptr(c.mean_old),
             ptr(c.L_old), 0, float(c.eps_mean), None, ptr(pm), None, None, N,
             K, st)
        pL = new(1, K, K)
        # one context buffer per update: the next epoch's eigen-decomposition
        # starts from this epoch's eigenvectors (the backward kernel below has
        # consumed the context by then)
        cbuf = c.proj_ctx
        call("tce_kl_cov_proj_fwd_" + s, ptr(L_new), ptr(c.L_old), 0,
             float(c.eps_cov), ptr(c.beta), c.entropy_eq, ptr(pL), ptr(cbuf),
             1, K, 1, st)
        # ---- pair log-prob of the stored trajectories under the projection
        logp = new(N, P)
        B, flag = ops._mp_ws(mp, T, dev)
        work = ops._pl_work(mean_new, N, P, mp, 0, True)
        flags = c.general | (ops._times_flags(mp, c.times, c.t0) & 2)
        pl = lambda f: (ptr(c.traj), ptr(pm), ptr(pL), 0, ptr(c.pairs),
                        *mp.c_args(), ptr(c.times), f, ptr(c.t0), ptr(c.y0),
                        ptr(c.v0), mp.cov_reg)
        pl_args = pl(flags)
        call("tce_pair_logprob_fwd_" + s, *pl_args, ptr(logp), ptr(B),
             ptr(flag), ptr(work), N, T, P, mp.num_dof, st)
        # ---- surrogate loss and d/d logp
        sur = new(2)
        glp = new(N, P)
        call("tce_surrogate_" + s, ptr(logp), ptr(c.lp_old), ptr(c.adv),
             N * P, ptr(sur), ptr(glp), ptr(c.sur_ws), st)
        g_pm, g_pL = new(N, K), new(K, K)
        pl_args = pl(c.general | 2 | 4)     # the forward call left table + pair factors
        call("tce_pair_logprob_bwd_" + s, *pl_args, ptr(glp), ptr(g_pm),
             ptr(g_pL), ptr(B), ptr(flag), ptr(work), N, T, P, mp.num_dof, st)
        # ---- KL diagnostics, entropy, trust region loss (+ its gradients)
        out = new(16)
        g_mean, g_L = new(N, K), new(K, K)
        ws = torch.empty(_lib.load().tce_kl_shared_ws_len(N),
                         dtype=torch.float64, device=dev)
        call("tce_kl_shared_" + s, ptr(mean_new), ptr(c.mean_old), ptr(pm),
             ptr(L_new), ptr(c.L_old), ptr(pL), N, K, float(c.tr_coeff),
             c.tr_include_cov, ptr(out), ptr(g_mean), ptr(g_L), ptr(ws), st)
        if c.ent_coef != 0.0:           # d(-coef * entropy(proj)) / d proj_L
            g_pL = g_pL - c.ent_coef * torch.diag(1.0 / pL[0].diagonal())
        # ---- back through the projection
        gm_p = new(N, K)
        call("tce_vec_env_" + s, 1, 1, ptr(mean_new), ptr(c.mean_old),
             ptr(c.L_old), 0, float(c.eps_mean), ptr(g_pm), None, ptr(gm_p),
             None, N, K, st)
        gL_p = new(1, K, K)
        call("tce_kl_cov_proj_bwd_" + s, ptr(L_new), ptr(c.L_old), 0, ptr(pL),
             ptr(cbuf), ptr(g_pL), ptr(gL_p), 1, K, st)
        g_mean.add_(gm_p)
        g_L.add_(gL_p[0])
        ctx.save_for_backward(g_mean, g_L)
        # {surrogate, entropy loss, trust region loss, total, entropy, kl x 12}
        ent = out[12]
        parts = torch.stack([sur[0], -c.ent_coef * ent, out[13]])
        rec = torch.cat([parts, parts.sum()[None], ent[None], out[:12]])
        ctx.mark_non_differentiable(rec)
        return rec[3].clone(), rec

    @staticmethod
    def backward(ctx, g, _):
        g_mean, g_L = ctx.saved_tensors
        return g_mean * g, g_L * g, None


def policy_objective(mean_new, L_new, context):
    """-> (policy_loss 0-dim (differentiable), record [17] = surrogate,
    entropy_loss, trust_region_loss, policy_loss, entropy, 12 KL means)."""
    base = ops.first_matrix(L_new)
    base = base if base.is_contiguous() else base.contiguous()
    mean_new = mean_new if mean_new.is_contiguous() else mean_new.contiguous()
    return _Objective.apply(mean_new, base, context)
