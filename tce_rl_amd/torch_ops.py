"""``torch.ops.tce_rl_amd.*``: the hot-path operators as ``torch.library`` custom
ops ("what a replacement must export"), for callers that want the
operators inside ``torch.compile`` / ``torch.export`` graphs or under
``torch.library.opcheck``.  Each op has ONE implementation -- the HIP kernels
behind libtce_hip.so, registered for device type "cuda" -- a fake (shape)
implementation for tracing, and autograd where the reference differentiates
through it.  There is no CPU implementation: on a CPU tensor the dispatcher
raises (no fallback, by contract).

The agent itself calls the same kernels through ``tce_rl_amd.ops`` directly
(no dispatcher hop on the ~30 launches of a policy epoch).

    gae(rewards, values, dones, time_limit_dones, gamma, lam, use_gae) -> (adv, ret)
        TemporalCorrelatedAgent.get_advantage_return, temporal_correlated_agent.py:118-181
    segment_advantage(mode, rewards, values, advantages, pred_pairs, gamma, norm, clip) -> [N,P]
        get_segment_advantage :183-321
    mdp_reward(step_rewards, event_flags) -> rewards        util_experiment.py:261-328
    rms_update(x, mean, var, count) -> ()  (mean / var in place)   util_numerical.py:315-337
    mvn_log_prob(x, mean, L) -> [N]   (+ backward)          black_box_policy.py:95-128
    maha(x, y, L) -> [N]   (+ backward)                     black_box_policy.py:205-224
    kl_mean_projection(mean, mean_old, L_old, eps) -> mean  (+ backward)   projection layer
    kl_cov_projection(L, L_old, eps_cov) -> L               (+ backward)   projection layer
    critic_values(x, w1, b1, w2, b2, w3, b3, act) -> [R]    util_nn.py:225-246 (128 x 2 fp32)
    prodmp_traj(mp, times, params, t0, y0, v0) -> [N,T,2 dof]              temporal_correlated_policy.py:74-102
    prodmp_pair_logprob(mp, traj, mean, L, times, t0, y0, v0, pairs) -> [N,P]  (+ backward)  :104-203
    mvn_rsample(mean, L, eps) -> [N,K]; mvn_entropy(L) -> [...]  (+ backward)  black_box_policy.py:58-154
    critic_epoch(x, ret, old, clip, w1..b3, act) -> (stats [2], grad [P])  temporal_correlated_agent.py:343-366
    adam_flat(param, grad, m, v, state, lr, b1, b2, eps, wd, clip, scale)  abstract_agent.py:62-82
    flat_grad_norm(grad, bound) -> [3]                       util_numerical.py:244-275
    allreduce_flat(flat, average)                            the sharded path's exchange (RCCL)
(``mp``: handle from ``mp_handle(prodmp)``.)
"""
import torch
from torch.library import custom_op, register_autograd

from . import ops as _ops

_NS = "tce_rl_amd"
_DEV = "cuda"


# ---- rollout post-processing ------------------------------------------------
@custom_op(_NS + "::gae", mutates_args=(), device_types=_DEV)
def gae(rewards: torch.Tensor, values: torch.Tensor, dones: torch.Tensor,
        time_limit_dones: torch.Tensor, gamma: float, lam: float,
        use_gae: bool) -> tuple[torch.Tensor, torch.Tensor]:
    adv, ret = _ops.gae(rewards, values, dones, time_limit_dones, gamma, lam,
                        use_gae)
    return adv, ret


@gae.register_fake
def _(rewards, values, dones, time_limit_dones, gamma, lam, use_gae):
    return torch.empty_like(rewards), torch.empty_like(rewards)


@custom_op(_NS + "::segment_advantage", mutates_args=(), device_types=_DEV)
def segment_advantage(mode: str, rewards: torch.Tensor, values: torch.Tensor,
                      advantages: torch.Tensor, pred_pairs: torch.Tensor,
                      gamma: float, norm_advantages: bool,
                      clip_advantages: float) -> torch.Tensor:
    out = _ops.segment_advantage(mode, rewards, values, advantages, pred_pairs,
                                 gamma, norm_advantages, clip_advantages)
    return out.clone() if out.data_ptr() in (rewards.data_ptr(),
                                             advantages.data_ptr()) else out


@segment_advantage.register_fake
def _(mode, rewards, values, advantages, pred_pairs, gamma, norm_advantages,
      clip_advantages):
    return rewards.new_empty(rewards.shape[0], pred_pairs.shape[0])


@custom_op(_NS + "::mdp_reward", mutates_args=(), device_types=_DEV)
def mdp_reward(step_rewards: torch.Tensor,
               event_flags: torch.Tensor) -> torch.Tensor:
    return _ops.mdp_reward(step_rewards, event_flags)


@mdp_reward.register_fake
def _(step_rewards, event_flags):
    return torch.empty_like(step_rewards)


@custom_op(_NS + "::rms_update", mutates_args=("mean", "var"),
           device_types=_DEV)
def rms_update(x: torch.Tensor, mean: torch.Tensor, var: torch.Tensor,
               count: float) -> None:
    _ops.rms_update(x, mean, var, count)


@rms_update.register_fake
def _(x, mean, var, count):
    return None


# ---- parameter-space Gaussian -----------------------------------------------
def _vec(mode, bwd, x, y, L, eps, g, want_gL=False):
    Lc, sL = _ops.split_L(L)
    return _ops._vec_env(mode, bwd, x.contiguous(), y.contiguous(), Lc, sL, eps,
                         g, want_gL=want_gL)


@custom_op(_NS + "::mvn_log_prob", mutates_args=(), device_types=_DEV)
def mvn_log_prob(x: torch.Tensor, mean: torch.Tensor,
                 L: torch.Tensor) -> torch.Tensor:
    return _vec(2, False, x, mean, L, 0.0, None)


@mvn_log_prob.register_fake
def _(x, mean, L):
    return x.new_empty(x.shape[0])


@custom_op(_NS + "::mvn_log_prob_bwd", mutates_args=(), device_types=_DEV)
def _mvn_log_prob_bwd(g: torch.Tensor, x: torch.Tensor, mean: torch.Tensor,
                      L: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    N, K = x.shape
    gmean, gL = _vec(2, True, x, mean, L, 0.0, g.contiguous(), want_gL=True)
    if L.dim() == 2:                        # shared factor: sum over the envs
        gL = _ops.sum_dim0(gL.reshape(N, K * K)).reshape(K, K)
    return gmean, gL.reshape(L.shape)


@_mvn_log_prob_bwd.register_fake
def _(g, x, mean, L):
    return torch.empty_like(mean), torch.empty_like(L)


def _mvn_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _mvn_backward(ctx, g):
    x, mean, L = ctx.saved_tensors
    gmean, gL = torch.ops.tce_rl_amd.mvn_log_prob_bwd(g, x, mean, L)
    return -gmean, gmean, gL          # d/dx = -d/dmean


register_autograd(_NS + "::mvn_log_prob", _mvn_backward,
                  setup_context=_mvn_setup)


@custom_op(_NS + "::maha", mutates_args=(), device_types=_DEV)
def maha(x: torch.Tensor, y: torch.Tensor, L: torch.Tensor) -> torch.Tensor:
    return _vec(0, False, x, y, L, 0.0, None)


@maha.register_fake
def _(x, y, L):
    return x.new_empty(x.shape[0])


@custom_op(_NS + "::maha_bwd", mutates_args=(), device_types=_DEV)
def _maha_bwd(g: torch.Tensor, x: torch.Tensor, y: torch.Tensor,
              L: torch.Tensor) -> torch.Tensor:
    return _vec(0, True, x, y, L, 0.0, g.contiguous())[0]


@_maha_bwd.register_fake
def _(g, x, y, L):
    return torch.empty_like(x)


def _maha_backward(ctx, g):
    x, y, L = ctx.saved_tensors
    gx = torch.ops.tce_rl_amd.maha_bwd(g, x, y, L)
    return gx, -gx, None              # L: a constant at every call site


register_autograd(_NS + "::maha", _maha_backward, setup_context=_mvn_setup)


# ---- KL trust-region projection ---------------------------------------------
@custom_op(_NS + "::kl_mean_projection", mutates_args=(), device_types=_DEV)
def kl_mean_projection(mean: torch.Tensor, mean_old: torch.Tensor,
                       L_old: torch.Tensor, eps: float) -> torch.Tensor:
    return _vec(1, False, mean, mean_old, L_old, eps, None)


@kl_mean_projection.register_fake
def _(mean, mean_old, L_old, eps):
    return torch.empty_like(mean)


@custom_op(_NS + "::kl_mean_projection_bwd", mutates_args=(),
           device_types=_DEV)
def _kl_mean_projection_bwd(g: torch.Tensor, mean: torch.Tensor,
                            mean_old: torch.Tensor, L_old: torch.Tensor,
                            eps: float) -> torch.Tensor:
    return _vec(1, True, mean, mean_old, L_old, eps, g.contiguous())[0]


@_kl_mean_projection_bwd.register_fake
def _(g, mean, mean_old, L_old, eps):
    return torch.empty_like(mean)


def _mp_setup(ctx, inputs, output):
    mean, mean_old, L_old, eps = inputs
    ctx.save_for_backward(mean, mean_old, L_old)
    ctx.eps = eps


def _mp_backward(ctx, g):
    mean, mean_old, L_old = ctx.saved_tensors
    return torch.ops.tce_rl_amd.kl_mean_projection_bwd(
        g, mean, mean_old, L_old, ctx.eps), None, None, None


register_autograd(_NS + "::kl_mean_projection", _mp_backward,
                  setup_context=_mp_setup)


@custom_op(_NS + "::kl_cov_projection", mutates_args=(), device_types=_DEV)
def kl_cov_projection(L: torch.Tensor, L_old: torch.Tensor,
                      eps_cov: float) -> tuple[torch.Tensor, torch.Tensor]:
    """-> (projected factors [B,K,K], context for the backward)."""
    from ._lib import call, load, ptr, sfx, stream
    L, Lo = L.contiguous(), L_old.contiguous()
    B, K = L.shape[0], L.shape[-1]
    sLo = 0 if Lo.dim() == 2 else K * K
    proj = torch.empty_like(L)
    cbuf = torch.empty(B, load().tce_kl_cov_proj_ctx_len(K),
                       dtype=torch.float64, device=L.device)
    call("tce_kl_cov_proj_fwd_" + sfx(L.dtype), ptr(L), ptr(Lo), sLo,
         float(eps_cov), None, 0, ptr(proj), ptr(cbuf), B, K, 0, stream())
    return proj, cbuf


@kl_cov_projection.register_fake
def _(L, L_old, eps_cov):
    from ._lib import load
    n = load().tce_kl_cov_proj_ctx_len(L.shape[-1])
    return torch.empty_like(L), L.new_empty(L.shape[0], n, dtype=torch.float64)


@custom_op(_NS + "::kl_cov_projection_bwd", mutates_args=(), device_types=_DEV)
def _kl_cov_projection_bwd(g: torch.Tensor, L: torch.Tensor,
                           L_old: torch.Tensor, proj: torch.Tensor,
                           ctxbuf: torch.Tensor) -> torch.Tensor:
    from ._lib import call, ptr, sfx, stream
    L, Lo = L.contiguous(), L_old.contiguous()
    B, K = L.shape[0], L.shape[-1]
    gL = torch.empty_like(L)
    call("tce_kl_cov_proj_bwd_" + sfx(L.dtype), ptr(L), ptr(Lo),
         0 if Lo.dim() == 2 else K * K, ptr(proj), ptr(ctxbuf),
         ptr(g.contiguous()), ptr(gL), B, K, stream())
    return gL


@_kl_cov_projection_bwd.register_fake
def _(g, L, L_old, proj, ctxbuf):
    return torch.empty_like(L)


def _cp_setup(ctx, inputs, output):
    L, L_old, _ = inputs
    proj, cbuf = output
    ctx.save_for_backward(L, L_old, proj, cbuf)


def _cp_backward(ctx, g, _g_ctx):
    L, L_old, proj, cbuf = ctx.saved_tensors
    return torch.ops.tce_rl_amd.kl_cov_projection_bwd(g, L, L_old, proj,
                                                      cbuf), None, None


register_autograd(_NS + "::kl_cov_projection", _cp_backward,
                  setup_context=_cp_setup)


# ---- fused critic forward ---------------------------------------------------
@custom_op(_NS + "::critic_values", mutates_args=(), device_types=_DEV)
def critic_values(x: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor,
                  w2: torch.Tensor, b2: torch.Tensor, w3: torch.Tensor,
                  b3: torch.Tensor, act: str) -> torch.Tensor:
    """V(x) [R] of the D_in <= 40 -> 128 -> 128 -> 1 float32 network
    (tce_mlp_critic_f32, forward only); x [R, D_in] (rows may be strided)."""
    from ._lib import call, ptr, stream
    from .critic_ops import _ACT
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32
    R, din = x.shape
    out = torch.empty(R, dtype=torch.float32, device=x.device)
    call("tce_mlp_critic_f32", ptr(x), 0, x.stride(0), R, R, din, ptr(w1),
         ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3), _ACT[act], None, None,
         0.0, ptr(out), None, None, None, 0, None, None, None, None, 0.0, 0.0,
         0.0, 0.0, 0.0, 0.0, 1.0, None, stream())
    return out


@critic_values.register_fake
def _(x, w1, b1, w2, b2, w3, b3, act):
    return x.new_empty(x.shape[0])


# ---- movement-primitive operators ---------------------------------------------
# The ProDMP's tables and scalars live in a Python object (tce_rl_amd.mp.ProDMP);
# an operator refers to it by a small integer handle (custom-op schemas carry
# tensors and scalars, not objects).
_MPS = {}


def mp_handle(mp) -> int:
    """Register a ProDMP for the operators below -> handle (stable per object)."""
    h = id(mp)
    _MPS[h] = mp
    return h


@custom_op(_NS + "::prodmp_traj", mutates_args=(), device_types=_DEV)
def prodmp_traj(mp: int, times: torch.Tensor, params: torch.Tensor,
                init_time: torch.Tensor, init_pos: torch.Tensor,
                init_vel: torch.Tensor) -> torch.Tensor:
    """cat[pos, vel] [N, T, 2 dof] of the ProDMP with parameters [N, K]
    (TemporalCorrelatedPolicy.sample, temporal_correlated_policy.py:74-102)."""
    return _ops.prodmp_traj(_MPS[mp], times, params, init_time, init_pos,
                            init_vel)


@prodmp_traj.register_fake
def _(mp, times, params, init_time, init_pos, init_vel):
    return params.new_empty(times.shape[0], times.shape[1],
                            2 * _MPS[mp].num_dof)


@custom_op(_NS + "::prodmp_pair_logprob", mutates_args=(), device_types=_DEV)
def prodmp_pair_logprob(mp: int, traj: torch.Tensor, mean: torch.Tensor,
                        L: torch.Tensor, times: torch.Tensor,
                        init_time: torch.Tensor, init_pos: torch.Tensor,
                        init_vel: torch.Tensor,
                        pred_pairs: torch.Tensor) -> torch.Tensor:
    """TemporalCorrelatedPolicy.log_prob -> [N, P]
    (temporal_correlated_policy.py:104-203); L: [K,K] (shared) or [N,K,K]."""
    with torch.no_grad():
        return _ops.pair_log_prob(_MPS[mp], traj, mean, L, times, init_time,
                                  init_pos, init_vel, pred_pairs)


@prodmp_pair_logprob.register_fake
def _(mp, traj, mean, L, times, init_time, init_pos, init_vel, pred_pairs):
    return mean.new_empty(mean.shape[0], pred_pairs.shape[0])


@custom_op(_NS + "::prodmp_pair_logprob_bwd", mutates_args=(),
           device_types=_DEV)
def _prodmp_pair_logprob_bwd(g: torch.Tensor, mp: int, traj: torch.Tensor,
                             mean: torch.Tensor, L: torch.Tensor,
                             times: torch.Tensor, init_time: torch.Tensor,
                             init_pos: torch.Tensor, init_vel: torch.Tensor,
                             pred_pairs: torch.Tensor
                             ) -> tuple[torch.Tensor, torch.Tensor]:
    from ._lib import call, ptr, sfx, stream
    m = _MPS[mp]
    c = lambda t: t if t.is_contiguous() else t.contiguous()
    Lc, sL = _ops.split_L(L)
    mean, traj, times, t0 = c(mean), c(traj), c(times), c(init_time)
    y0, v0, pairs = c(init_pos), c(init_vel), c(pred_pairs.to(torch.int64))
    N, T = times.shape
    P, K = pairs.shape[0], mean.shape[1]
    gmean = torch.empty_like(mean)
    gL = torch.empty((K, K) if sL == 0 else (N, K, K), dtype=mean.dtype,
                     device=mean.device)
    B, flag = _ops._mp_ws(m, T, mean.device)
    work = _ops._pl_work(mean, N, P, m, sL, True)
    general = 1 | (_ops._times_flags(m, times, t0) & 2)
    call("tce_pair_logprob_bwd_" + sfx(mean.dtype), ptr(traj), ptr(mean),
         ptr(Lc), sL, ptr(pairs), *m.c_args(), ptr(times), general, ptr(t0),
         ptr(y0), ptr(v0), m.cov_reg, ptr(c(g)), ptr(gmean), ptr(gL), ptr(B),
         ptr(flag), ptr(work), N, T, P, m.num_dof, stream())
    return gmean, gL.reshape(L.shape)


@_prodmp_pair_logprob_bwd.register_fake
def _(g, mp, traj, mean, L, times, init_time, init_pos, init_vel, pred_pairs):
    return torch.empty_like(mean), torch.empty_like(L)


def _pl_setup(ctx, inputs, output):
    mp, traj, mean, L, times, t0, y0, v0, pairs = inputs
    ctx.mp = mp
    ctx.save_for_backward(traj, mean, L, times, t0, y0, v0, pairs)


def _pl_backward(ctx, g):
    traj, mean, L, times, t0, y0, v0, pairs = ctx.saved_tensors
    gmean, gL = torch.ops.tce_rl_amd.prodmp_pair_logprob_bwd(
        g, ctx.mp, traj, mean, L, times, t0, y0, v0, pairs)
    return None, None, gmean, gL, None, None, None, None, None


register_autograd(_NS + "::prodmp_pair_logprob", _pl_backward,
                  setup_context=_pl_setup)


# ---- parameter-space sampling / entropy ---------------------------------------
@custom_op(_NS + "::mvn_rsample", mutates_args=(), device_types=_DEV)
def mvn_rsample(mean: torch.Tensor, L: torch.Tensor,
                eps: torch.Tensor) -> torch.Tensor:
    """mean + L eps (BlackBoxPolicy.sample, black_box_policy.py:58-93; the
    noise is an argument)."""
    return _ops.mvn_rsample(mean, L, eps)


@mvn_rsample.register_fake
def _(mean, L, eps):
    return torch.empty_like(mean)


@custom_op(_NS + "::mvn_entropy", mutates_args=(), device_types=_DEV)
def mvn_entropy(L: torch.Tensor) -> torch.Tensor:
    """K/2 (1 + log 2 pi) + sum log diag L per factor
    (BlackBoxPolicy.entropy, black_box_policy.py:149-154): [] for one [K,K]
    factor, [B] for [B,K,K]."""
    return _ops.mvn_entropy(L.detach()).clone()


@mvn_entropy.register_fake
def _(L):
    return L.new_empty(L.shape[:-2])


def _ent_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0])


def _ent_backward(ctx, g):
    L, = ctx.saved_tensors                      # d/dL = diag(1 / L_ii)
    d = torch.diag_embed(1.0 / L.diagonal(dim1=-2, dim2=-1))
    return g[..., None, None] * d


register_autograd(_NS + "::mvn_entropy", _ent_backward,
                  setup_context=_ent_setup)


# ---- fused critic epoch, flat Adam, gradient norm -----------------------------
@custom_op(_NS + "::critic_epoch", mutates_args=(), device_types=_DEV)
def critic_epoch(x: torch.Tensor, returns: torch.Tensor,
                 old_values: torch.Tensor, clip: float, w1: torch.Tensor,
                 b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor,
                 w3: torch.Tensor, b3: torch.Tensor,
                 act: str) -> tuple[torch.Tensor, torch.Tensor]:
    """One full-batch critic epoch of the D_in <= 40 -> 128 -> 128 -> 1 float32
    value network in ONE launch (+ the slab reduction): forward, (clipped)
    value loss, backward (update_critic's loop body,
    temporal_correlated_agent.py:343-366) -> (stats [2] = {mean loss,
    |grad|^2}, flat gradient [P] in the order W1, b1, W2, b2, w3, b3).
    x [R, D_in] (rows may be strided)."""
    from ._lib import call, load, ptr, stream
    from .critic_ops import _ACT
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32
    R, din = x.shape
    lib = load()
    P = lib.tce_mlp_critic_num_params(din)
    dev = x.device
    partials = torch.empty(lib.tce_mlp_critic_grid(), P + 2,
                           dtype=torch.float32, device=dev)
    grad = torch.empty(P, dtype=torch.float32, device=dev)
    stats = torch.zeros(2, dtype=torch.float32, device=dev)
    c = lambda t: t if t.is_contiguous() else t.contiguous()
    call("tce_mlp_critic_f32", ptr(x), 0, x.stride(0), R, R, din, ptr(c(w1)),
         ptr(c(b1)), ptr(c(w2)), ptr(c(b2)), ptr(c(w3)), ptr(c(b3)), _ACT[act],
         ptr(c(returns)), ptr(c(old_values)) if clip > 0 else None,
         float(clip), None, ptr(partials), ptr(grad), ptr(stats), 0, None,
         None, None, None, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, None, stream())
    return stats, grad


@critic_epoch.register_fake
def _(x, returns, old_values, clip, w1, b1, w2, b2, w3, b3, act):
    P = w1.numel() + b1.numel() + w2.numel() + b2.numel() + w3.numel() + \
        b3.numel()
    return x.new_empty(2), x.new_empty(P)


@custom_op(_NS + "::adam_flat", mutates_args=("param", "m", "v", "state"),
           device_types=_DEV)
def adam_flat(param: torch.Tensor, grad: torch.Tensor, m: torch.Tensor,
              v: torch.Tensor, state: torch.Tensor, lr: float, beta1: float,
              beta2: float, eps: float, weight_decay: float, clip: float,
              grad_scale: float) -> None:
    """grad_norm_clip (util_numerical.py:244-275) + the Adam step of
    torch.optim.Adam(lr, weight_decay) (abstract_agent.py:62-82) on flat
    buffers; state = {step, |g|, |g| clipped, factor} stays on the device."""
    from ._lib import call, ptr, sfx, stream
    call("tce_adam_flat_" + sfx(param.dtype), ptr(param), ptr(grad), ptr(m),
         ptr(v), param.numel(), ptr(state), None, lr, beta1, beta2, eps,
         weight_decay, clip, grad_scale, stream())


@adam_flat.register_fake
def _(param, grad, m, v, state, lr, beta1, beta2, eps, weight_decay, clip,
      grad_scale):
    return None


@custom_op(_NS + "::flat_grad_norm", mutates_args=(), device_types=_DEV)
def flat_grad_norm(grad: torch.Tensor, bound: float) -> torch.Tensor:
    """{|g|, |g| after clipping to `bound` (<= 0: none), clip factor} [3] of a
    flat gradient, without touching it (the two norms grad_norm_clip reports,
    util_numerical.py:244-275), no host sync."""
    n = grad.reshape(-1).norm(2)
    coef = torch.clamp(bound / (n + 1e-6), max=1.0) if bound > 0 \
        else torch.ones_like(n)
    return torch.stack([n, n * coef, coef])


@flat_grad_norm.register_fake
def _(grad, bound):
    return grad.new_empty(3)


@custom_op(_NS + "::allreduce_flat", mutates_args=("flat",),
           device_types=_DEV)
def allreduce_flat(flat: torch.Tensor, average: bool) -> None:
    """In-place sum (mean) of a flat gradient buffer over the ranks of the
    default process group (RCCL); identity without one.  The env-sharded
    path's one exchange per optimizer step."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(flat)
        if average:
            flat.div_(dist.get_world_size())


@allreduce_flat.register_fake
def _(flat, average):
    return None
