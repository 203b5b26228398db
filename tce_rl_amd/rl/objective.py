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
            and not proj.entropy_first
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
        # all segments start at the same time (every shipped config): checked
        # once per update on the host (one small copy), so the ~100 objective
        # evaluations skip the launches of the general-path kernels
        if not self.general and bool((self.t0 == self.t0.flatten()[0]).all()):
            self.general |= 8
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
        self.ws = self.kl_ws = self.pl_work = self.ws_key = None


def bb_supported(agent, L_old):
    """The fused objective of the black-box agent: a shared (non-contextual)
    covariance under the KL projection."""
    pol, proj = agent.policy, agent.projection
    return (not pol.contextual_std and type(proj) is KLProjectionLayer
            and not proj.entropy_first
            and ops.split_L(L_old)[1] == 0)


class BBContext:
    """Per-update constants of the black-box agent's objective
    (black_box_agent.py:283-357): old distribution, sampled parameter vectors,
    their old log-probs and advantages.  The tensors are used as given (the
    agent passes its static buffers when it keeps the epoch graph)."""
    bb = True

    def __init__(self, agent, mean_old, L_old, actions, lp_old, adv, beta):
        c = lambda t: t if t.is_contiguous() else t.contiguous()
        pol, proj = agent.policy, agent.projection
        self.mean_old, self.actions = c(mean_old), c(actions)
        self.L_old = c(ops.split_L(L_old)[0].detach())
        self.lp_old, self.adv = c(lp_old), c(adv)
        self.eps_mean, self.eps_cov = proj.mean_bound, proj.cov_bound
        self.beta = None if beta is None else \
            c(beta.detach().to(self.mean_old.dtype).reshape(1))
        self.entropy_eq = int(bool(proj.entropy_eq))
        self.tr_coeff = proj.trust_region_coeff
        self.tr_include_cov = int(pol.contextual_std or not agent.set_variance)
        self.ent_coef = float(agent.entropy_penalty_coef)
        N, K = self.mean_old.shape
        dev, dt, lib = self.mean_old.device, self.mean_old.dtype, _lib.load()
        f64 = torch.float64
        self.proj_ctx = torch.zeros(lib.tce_kl_cov_proj_ctx_len(K), dtype=f64,
                                    device=dev)
        self.sur_ws = torch.zeros(lib.tce_surrogate_ws_len(), dtype=f64,
                                  device=dev)
        self.kl_ws = torch.empty(lib.tce_kl_shared_ws_len(N), dtype=f64,
                                 device=dev)
        self.ws = torch.empty(lib.tce_bb_policy_objective_ws_len(N, K),
                              dtype=dt, device=dev)
        # the projected distribution of the latest evaluation
        self.proj_mean = torch.empty(N, K, dtype=dt, device=dev)
        self.proj_L = torch.empty(K, K, dtype=dt, device=dev)


def evaluate_bb(mean_new, L_new, c):
    """evaluate() for the black-box agent: ONE C call
    (tce_bb_policy_objective_*)."""
    N, K = mean_new.shape
    dt, dev = mean_new.dtype, mean_new.device
    new = lambda *shape: torch.empty(*shape, dtype=dt, device=dev)
    g_mean, g_L, sur, out = new(N, K), new(K, K), new(2), new(16)
    call("tce_bb_policy_objective_" + sfx(dt), ptr(mean_new), ptr(L_new),
         ptr(c.mean_old), ptr(c.L_old), ptr(c.actions), ptr(c.lp_old),
         ptr(c.adv), float(c.eps_mean), float(c.eps_cov), ptr(c.beta),
         c.entropy_eq, ptr(c.proj_ctx), float(c.tr_coeff), c.tr_include_cov,
         c.ent_coef, ptr(c.sur_ws), ptr(c.kl_ws), ptr(c.ws), ptr(g_mean),
         ptr(g_L), ptr(sur), ptr(out), ptr(c.proj_mean), ptr(c.proj_L), N, K,
         stream())
    return g_mean, g_L, sur, out


def _workspaces(c, N, K, P, ref):
    lib = _lib.load()
    if c.ws is None or c.ws_key != (N, K, P, ref.dtype):
        c.ws = torch.empty(lib.tce_policy_objective_ws_len(N, K, P),
                           dtype=ref.dtype, device=ref.device)
        c.kl_ws = torch.empty(lib.tce_kl_shared_ws_len(N),
                              dtype=torch.float64, device=ref.device)
        c.pl_work = ops._pl_work(ref, N, P, c.mp, 0, True)
        c.ws_key = (N, K, P, ref.dtype)


def begin(var_vec, min_std, L_out, c, N):
    """Cholesky head + covariance projection of the coming ``evaluate(...,
    started=True)`` on the library's second stream (tce_policy_objective_begin_*):
    they only need the variance parameters and run beside the mean net."""
    K, P = L_out.shape[-1], c.pairs.shape[0]
    _workspaces(c, N, K, P, L_out)
    call("tce_policy_objective_begin_" + sfx(L_out.dtype), ptr(var_vec),
         var_vec.numel(), float(min_std), ptr(c.L_old), float(c.eps_cov),
         ptr(c.beta), c.entropy_eq, ptr(c.proj_ctx), ptr(L_out), ptr(c.ws), N,
         K, P, stream())


def end(g_L, c, N):
    """Join of a deferred ``evaluate`` (tce_policy_objective_end_*): adds the
    covariance projection's part to g_L."""
    K, P = g_L.shape[-1], c.pairs.shape[0]
    call("tce_policy_objective_end_" + sfx(g_L.dtype), ptr(g_L), ptr(c.ws), N,
         K, P, stream())


def evaluate(mean_new, L_new, c, started=False, defer=False):
    """The objective of one epoch and its gradient, kernels only (no autograd):
    -> (g_mean [N, K], g_L [K, K], sur [2], out [16]) with sur[0] the surrogate
    loss and out = {12 KL means, entropy, trust region loss, -, -}.  ONE C call
    (tce_policy_objective_*): ~14 kernels, the single-workgroup K x K ones on
    a second stream beside the per-env ones."""
    N, K = mean_new.shape
    dt, dev = mean_new.dtype, mean_new.device
    mp = c.mp
    T, P = c.times.shape[1], c.pairs.shape[0]
    _workspaces(c, N, K, P, mean_new)
    new = lambda *shape: torch.empty(*shape, dtype=dt, device=dev)
    g_mean, g_L, sur, out = new(N, K), new(K, K), new(2), new(16)
    B, flag = ops._mp_ws(mp, T, dev)
    flags = c.general | (ops._times_flags(mp, c.times, c.t0) & 2)
    # one context buffer per update: the next epoch's eigen-decomposition
    # starts from this epoch's eigenvectors
    call("tce_policy_objective_" + sfx(dt), ptr(mean_new), ptr(L_new),
         ptr(c.mean_old), ptr(c.L_old), ptr(c.traj), ptr(c.lp_old), ptr(c.adv),
         ptr(c.pairs), *mp.c_args(), ptr(c.times), flags, c.general | 2 | 4,
         ptr(c.t0), ptr(c.y0), ptr(c.v0), mp.cov_reg, ptr(B), ptr(flag),
         ptr(c.pl_work), float(c.eps_mean), float(c.eps_cov), ptr(c.beta),
         c.entropy_eq, ptr(c.proj_ctx), float(c.tr_coeff), c.tr_include_cov,
         c.ent_coef, ptr(c.sur_ws), ptr(c.kl_ws), ptr(c.ws), ptr(g_mean),
         ptr(g_L), ptr(sur), ptr(out), N, T, P, mp.num_dof, K, int(started),
         int(defer), stream())
    return g_mean, g_L, sur, out


class _Objective(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mean_new, L_new, c):
        g_mean, g_L, sur, out = evaluate_bb(mean_new, L_new, c) \
            if getattr(c, "bb", False) else evaluate(mean_new, L_new, c)
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


class DirectEpoch:
    """One policy epoch of the fused objective WITHOUT autograd, ONE C call
    (tce_policy_epoch2_*): forward of the mean net and of the Cholesky head,
    the objective and its gradient, then the parameter gradients written
    straight into the optimizer's flat gradient buffer, the Cholesky head's
    backward, flat Adam and the record row.  Same arithmetic as the autograd
    path (``policy_objective`` + ``backward()``; the parity tests run both),
    ~26 launches per epoch instead of ~56.

    Mean nets (``kind``): 0 = float32 D_in <= 40 -> 128 -> 128 -> K on the fused
    MFMA kernels of csrc/mlp.hip + the output layer's row kernel; 1 = the row
    kernels of csrc/pmlp.hip (float32 / float64, one or two hidden layers:
    box pushing's float64 128 x 2 net, table tennis's 256 x 1 tanh net).
    Parameters in a ``FlatAdam`` in the order mean net, variance variable.

    ``run(row, balance=True)``: the epoch of a balance-check iteration
    (temporal_correlated_agent.py:447-522) -- the gradient norms of the
    surrogate loss alone and of the trust region loss alone come from ONE
    evaluation of the objective whose gradient is kept in two parts."""

    @staticmethod
    def kind(agent, states):
        import os
        from .. import critic_ops, pmlp_ops
        from ..optim import FlatAdam
        pol, opt = agent.policy, agent.policy_optimizer
        net = pol.mean_net
        if not isinstance(opt, FlatAdam) or pol.contextual_cov or \
                states.dim() != 2 or net.act_func_last_type is not None or \
                states.dtype != net.dtype:
            return None
        params = list(net.parameters()) + [pol.variance_net.variable]
        if len(params) != len(opt._params) or \
                not all(a is b for a, b in zip(params, opt._params)):
            return None
        force = os.environ.get("TCE_POLICY_NET", "")
        if critic_ops.hidden_supported(net, states) and force != "pmlp":
            return 0
        if pmlp_ops.supported(net) and opt.flat_param.data_ptr() % 16 == 0 \
                and force != "library":
            return 1
        return None

    @staticmethod
    def supported(agent, states):
        return DirectEpoch.kind(agent, states) is not None

    def __init__(self, agent, states, context):
        from .. import critic_ops, pmlp_ops
        self.agent, self.c = agent, context
        pol = agent.policy
        self.net, self.opt = pol.mean_net, agent.policy_optimizer
        self.net_kind = DirectEpoch.kind(agent, states)
        self.var = pol.variance_net.variable
        self.x = states if states.is_contiguous() else states.contiguous()
        self.N, self.din = self.x.shape
        self.K, self.min_std = pol.dim_out, float(pol.min_std)
        self.act = critic_ops._ACT[self.net.act_func_hidden_type]
        dev, dt, lib = self.x.device, self.x.dtype, _lib.load()
        self.sfx = sfx(dt)
        hl = list(self.net.hidden_layers)
        self.H, self.NL = hl[0], len(hl)
        flat = self.opt.flat_grad
        self.nvec = self.var.numel()
        nparam = flat.numel()
        if self.net_kind == 0:
            P = lib.tce_mlp_critic_num_params(self.din)
            assert nparam == 128 * self.din + 128 + 128 * 128 + 128 + \
                self.K * 128 + self.K + self.nvec
            self.partials = torch.empty(min(lib.tce_mlp_critic_grid(),
                                            (self.N + 63) // 64), P + 2,
                                        dtype=dt, device=dev)
            self.ol_ws = torch.empty(
                lib.tce_out_layer_grad_ws_len(self.N, self.K, 128), dtype=dt,
                device=dev)
        else:
            P = lib.tce_pmlp_num_params(self.din, self.H, self.NL, self.K)
            assert nparam == P + self.nvec
            self.partials = torch.empty(lib.tce_pmlp_max_slabs() * P, dtype=dt,
                                        device=dev)
            self.ol_ws = None
        # workspace of tce_policy_epoch2_*: h2 | h1 (gh) | mean | g_mean | L |
        # g_L | - | sur2 [4] | out16 [16] | stats [12] | a parameter gradient
        self.ws = torch.zeros(lib.tce_policy_epoch2_ws_len(self.N, self.K, self.H,
                                                           nparam),
                              dtype=dt, device=dev)
        up4 = lambda n: (n + 3) // 4 * 4
        o = 2 * up4(self.N * self.H) + 2 * up4(self.N * self.K) + \
            3 * up4(self.K * self.K)
        self.sur, self.out16 = self.ws[o:o + 2], self.ws[o + 4:o + 20]

    def run(self, rec_row, balance=False, bal=None):
        """One epoch; rec_row [19] receives {surrogate, entropy loss, trust
        region loss, total, entropy, |g|, |g| clipped, 12 KL means}; balance:
        bal [2] receives the gradient norms of the surrogate / trust region
        loss alone.  Env shards: the epoch's last launch adds the peers'
        gradients (``agent.xchg_policy``) -- still one call; without an
        in-library exchange the call stops in front of the optimizer step, the
        flat gradient is all-reduced, then stepped and recorded."""
        c, ag, opt = self.c, self.agent, self.opt
        x, N, K = self.x, self.N, self.K
        mp = c.mp
        T, P = c.times.shape[1], c.pairs.shape[0]
        _workspaces(c, N, K, P, x)
        B, flag = ops._mp_ws(mp, T, x.device)
        flags = c.general | (ops._times_flags(mp, c.times, c.t0) & 2)
        opt.bind_grads()
        assert rec_row.is_contiguous() and rec_row.numel() == 19
        g = opt.param_groups[0]
        # env shards: the gradient exchange rides in the epoch's last launch
        # (ag.xchg_policy); without it the call stops in front of the step
        xch = ag.xchg_policy if ag.dist.active else None
        do_adam = not ag.dist.active or xch is not None
        gscale = 1.0 / ag.dist.world if xch is not None else 1.0
        assert do_adam or not balance
        if do_adam:
            opt.host_step += 1
            opt._opt_called = True            # for LinearLR's order check
        call("tce_policy_epoch2_" + self.sfx, ptr(x), x.stride(0), N, self.din,
             self.H, self.NL, self.net_kind, self.act, self.nvec, self.min_std,
             ptr(opt.flat_param), ptr(opt.flat_grad), ptr(c.mean_old),
             ptr(c.L_old), ptr(c.traj), ptr(c.lp_old), ptr(c.adv), ptr(c.pairs),
             *mp.c_args(), ptr(c.times), flags, c.general | 2 | 4, ptr(c.t0),
             ptr(c.y0), ptr(c.v0), mp.cov_reg, ptr(B), ptr(flag),
             ptr(c.pl_work), float(c.eps_mean), float(c.eps_cov), ptr(c.beta),
             c.entropy_eq, ptr(c.proj_ctx), float(c.tr_coeff), c.tr_include_cov,
             c.ent_coef, ptr(c.sur_ws), ptr(c.kl_ws), ptr(c.ws), ptr(self.ws),
             ptr(self.partials), ptr(self.ol_ws), T, P, mp.num_dof, K,
             ptr(opt.m), ptr(opt.v), ptr(opt.dev_state), float(g["lr"]),
             float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
             float(g["weight_decay"]), float(ag.clip_grad_norm), gscale,
             int(do_adam), int(balance), ptr(rec_row), ptr(bal),
             None if xch is None else xch.handle, stream())
        if not do_adam:
            ag._optimizer_step(opt, ag.policy_net_params, ag.clip_grad_norm,
                               want_norms=False)
            call("tce_policy_record_" + self.sfx, ptr(self.sur), ptr(self.out16),
                 ptr(opt.dev_state) + opt.dev_state.element_size(),
                 c.ent_coef, ptr(rec_row), stream())


class BBDirectEpoch:
    """One policy epoch of the black-box agent WITHOUT autograd, ONE C call
    (tce_bb_policy_epoch_*; black_box_agent.py:225-339), for the mean nets of
    ``DirectEpoch`` -- the reference's box-pushing (128 x 2) and table-tennis
    (256 x 1) BBRL policies, which the 64-wide row kernels of csrc/smlp.hip do
    not cover: Cholesky head + covariance projection beside the mean net's
    forward, the black-box objective and its gradient, the net's backward into
    the optimizer's flat gradient buffer, Cholesky head backward + clip + Adam +
    record row as one launch.  ``balance``: the epoch of a balance-check
    iteration (:226-284), the two gradient norms from ONE evaluation whose
    gradient is kept in two parts."""

    @staticmethod
    def supported(agent, states, L_old):
        return bb_supported(agent, L_old) and \
            DirectEpoch.kind(agent, states) is not None

    def __init__(self, agent, states, context):
        from .. import critic_ops
        self.agent, self.c = agent, context
        pol = agent.policy
        self.net, self.opt = pol.mean_net, agent.policy_optimizer
        self.net_kind = DirectEpoch.kind(agent, states)
        self.var = pol.variance_net.variable
        self.x = states if states.is_contiguous() else states.contiguous()
        self.N, self.din = self.x.shape
        self.K, self.min_std = pol.dim_out, float(pol.min_std)
        self.act = critic_ops._ACT[self.net.act_func_hidden_type]
        dev, dt, lib = self.x.device, self.x.dtype, _lib.load()
        self.sfx = sfx(dt)
        hl = list(self.net.hidden_layers)
        self.H, self.NL = hl[0], len(hl)
        self.nvec = self.var.numel()
        nparam = self.opt.flat_grad.numel()
        # scratch kept on the net across updates (same shapes every iteration)
        cache = self.net.__dict__.setdefault("_tce_bb_direct", {})
        key = (self.N, self.K, self.net_kind, dt)
        bufs = cache.get(key)
        if bufs is None:
            if self.net_kind == 0:
                P = lib.tce_mlp_critic_num_params(self.din)
                partials = torch.empty(min(lib.tce_mlp_critic_grid(),
                                           (self.N + 63) // 64), P + 2,
                                       dtype=dt, device=dev)
                ol_ws = torch.empty(
                    lib.tce_out_layer_grad_ws_len(self.N, self.K, 128),
                    dtype=dt, device=dev)
            else:
                P = lib.tce_pmlp_num_params(self.din, self.H, self.NL, self.K)
                assert nparam == P + self.nvec
                partials = torch.empty(lib.tce_pmlp_max_slabs() * P, dtype=dt,
                                       device=dev)
                ol_ws = None
            # (zeroed once: the tail kernel re-arms its ticket)
            ws = torch.zeros(lib.tce_policy_epoch2_ws_len(
                self.N, self.K, self.H, nparam), dtype=dt, device=dev)
            bufs = cache[key] = (partials, ol_ws, ws)
            for k in [k for k in cache if k != key]:
                del cache[k]
        self.partials, self.ol_ws, self.ws = bufs
        up4 = lambda n: (n + 3) // 4 * 4
        self._o_mean = 2 * up4(self.N * self.H)
        self._o_L = self._o_mean + 2 * up4(self.N * self.K)
        o = self._o_L + 3 * up4(self.K * self.K)
        self.sur, self.out16 = self.ws[o:o + 2], self.ws[o + 4:o + 20]

    def latest(self):
        """(mean_new [N,K], L_new [K,K]) of the latest epoch (copies)."""
        N, K = self.N, self.K
        return (self.ws[self._o_mean:self._o_mean + N * K].view(N, K).clone(),
                self.ws[self._o_L:self._o_L + K * K].view(K, K).clone())

    def run(self, rec_row, balance=False, bal=None, last=False):
        """One epoch; rec_row [19] as DirectEpoch.run; last: the projected
        distribution is left in the context (proj_mean / proj_L)."""
        c, ag, opt = self.c, self.agent, self.opt
        x, N, K = self.x, self.N, self.K
        opt.bind_grads()
        assert rec_row.is_contiguous() and rec_row.numel() == 19
        g = opt.param_groups[0]
        xch = ag.xchg_policy if ag.dist.active else None
        do_adam = not ag.dist.active or xch is not None
        gscale = 1.0 / ag.dist.world if xch is not None else 1.0
        assert do_adam or not balance
        if do_adam:
            opt.host_step += 1
            opt._opt_called = True            # for LinearLR's order check
        call("tce_bb_policy_epoch_" + self.sfx, ptr(x), x.stride(0), N, self.din,
             self.H, self.NL, self.net_kind, self.act, self.nvec, self.min_std,
             ptr(opt.flat_param), ptr(opt.flat_grad), ptr(c.mean_old),
             ptr(c.L_old), ptr(c.actions), ptr(c.lp_old), ptr(c.adv),
             float(c.eps_mean), float(c.eps_cov), ptr(c.beta), c.entropy_eq,
             ptr(c.proj_ctx), float(c.tr_coeff), c.tr_include_cov, c.ent_coef,
             ptr(c.sur_ws), ptr(c.kl_ws), ptr(c.ws), ptr(self.ws),
             ptr(self.partials), ptr(self.ol_ws), K, ptr(opt.m), ptr(opt.v),
             ptr(opt.dev_state), float(g["lr"]), float(g["betas"][0]),
             float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
             float(ag.clip_grad_norm), gscale, int(do_adam), int(balance),
             ptr(rec_row), ptr(bal), ptr(c.proj_mean) if last else None,
             ptr(c.proj_L) if last else None,
             None if xch is None else xch.handle, stream())
        if not do_adam:
            ag.dist.allreduce_flat(opt.flat_grad, ag._policy_group,
                                   average=False)
            opt.step_once(ag.clip_grad_norm, grad_scale=1.0 / ag.dist.world)
            call("tce_policy_record_" + self.sfx, ptr(self.sur), ptr(self.out16),
                 ptr(opt.dev_state) + opt.dev_state.element_size(),
                 c.ent_coef, ptr(rec_row), stream())
