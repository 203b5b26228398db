"""Torch-tensor front end of the HIP kernels (libtce_hip.so).

Every function takes HIP device tensors, enqueues kernels on the current
stream through the C ABI (include/tce_hip.h) and returns device tensors.
PyTorch is used for memory and streams only.  No CPU fallback.
"""
import torch

from . import _lib
from ._lib import call, ptr, stream, sfx, check_dev


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _bool_u8(t):
    """bool / uint8 tensor -> contiguous 1-byte view."""
    t = _c(t)
    if t.dtype == torch.bool:
        return t.view(torch.uint8)
    assert t.dtype == torch.uint8
    return t


# ---------------------------------------------------------------------------
# moments / normalisation
# ---------------------------------------------------------------------------
def merge_stats(stats, group=None):
    """All-gather (count, mean, M2) over the ranks of `group` and merge them
    (exact pooled-moments identity), so that a sharded batch normalises with
    the statistics of the global batch."""
    import torch.distributed as dist
    from .dist import active, all_gather_into_tensor
    if not active(group):
        return stats
    w = dist.get_world_size(group)
    buf = stats.new_empty(w, 3)
    all_gather_into_tensor(buf, stats.reshape(1, 3), group=group)
    if buf.is_cuda and buf.dtype == torch.float64:
        # the ranks' (count, mean, M2) triples are partials like any others:
        # the kernel that merges a launch's partials merges them (one launch)
        out = torch.empty(3, dtype=torch.float64, device=buf.device)
        call("tce_moments_finalize", ptr(buf), w, ptr(out), stream())
        return out
    n = buf[:, 0].sum()
    mean = (buf[:, 0] * buf[:, 1]).sum() / n
    m2 = (buf[:, 2] + buf[:, 0] * (buf[:, 1] - mean) ** 2).sum()
    return torch.stack([n, mean, m2])


def _finalize(partials, group=None):
    stats = torch.empty(3, dtype=torch.float64, device=partials.device)
    call("tce_moments_finalize", ptr(partials), partials.shape[0], ptr(stats),
         stream())
    return merge_stats(stats, group)


def moments(x, group=None):
    """{count, mean, M2} (float64[3], device) of all elements of x."""
    check_dev(x)
    x = _c(x)
    nparts = _lib.load().tce_moments_num_partials()
    partials = torch.empty(nparts, 3, dtype=torch.float64, device=x.device)
    call("tce_moments_partial_" + sfx(x.dtype), ptr(x), x.numel(),
         ptr(partials), stream())
    return _finalize(partials, group)


def normalize(x, stats=None, eps=1e-8, clip=0.0, single_std_one=False):
    """clamp((x - mean) / (std_unbiased + eps), +-clip)."""
    check_dev(x)
    x = _c(x)
    y = torch.empty_like(x)
    call("tce_normalize_" + sfx(x.dtype), ptr(x), ptr(y), x.numel(),
         ptr(stats), float(eps), float(clip), int(single_std_one), stream())
    return y


# ---------------------------------------------------------------------------
# GAE + segment advantage
# ---------------------------------------------------------------------------
def gae(rewards, values, dones, time_limit_dones, gamma, lam, use_gae=True,
        pred_pairs=None):
    """TemporalCorrelatedAgent.get_advantage_return on the GPU.

    Returns (adv, ret) or, with pred_pairs [P,2] int64, (adv, ret, seg_raw,
    partials): the un-normalised value_subtraction segment advantage [N,P] and
    the per-workgroup moment partials for its global normalisation."""
    check_dev(rewards, values, dones, time_limit_dones, pred_pairs)
    N, T = rewards.shape
    assert values.shape == (N, T + 1) and dones.shape == (N, T)
    assert values.dtype == rewards.dtype
    r, v = _c(rewards), _c(values)
    d, tl = _bool_u8(dones), _bool_u8(time_limit_dones)
    adv, ret = torch.empty_like(r), torch.empty_like(r)
    P, seg, partials, pairs = 0, None, None, None
    if pred_pairs is not None:
        pairs = _c(pred_pairs.to(torch.int64))
        P = pairs.shape[0]
        seg = torch.empty(N, P, dtype=r.dtype, device=r.device)
        partials = torch.empty(_lib.load().tce_gae_num_partials(N), 3,
                               dtype=torch.float64, device=r.device)
    call("tce_gae_" + sfx(r.dtype), ptr(r), ptr(v), ptr(d), ptr(tl), ptr(adv),
         ptr(ret), ptr(pairs), P, ptr(seg), ptr(partials), N, T, float(gamma),
         float(lam), int(bool(use_gae)), stream())
    if pred_pairs is None:
        return adv, ret
    return adv, ret, seg, partials


def segment_advantage(mode, rewards, values, advantages, pred_pairs, gamma,
                      norm_advantages=False, clip_advantages=0.0, group=None,
                      fused=None):
    """TemporalCorrelatedAgent.get_segment_advantage on the GPU.

    fused = (seg_raw, partials) from gae(..., pred_pairs) skips recomputing the
    value_subtraction sums."""
    check_dev(rewards, values, advantages, pred_pairs)
    N, T = rewards.shape
    pairs = _c(pred_pairs.to(torch.int64))
    P = pairs.shape[0]
    s = sfx(rewards.dtype)
    if mode == "value_subtraction":
        if fused is None:
            z = torch.zeros(N, T, dtype=torch.bool, device=rewards.device)
            _, _, seg, partials = gae(rewards, values, z, z, gamma, 0.0, True,
                                      pairs)
        else:
            seg, partials = fused
        if not norm_advantages:
            return seg
        return normalize(seg, _finalize(partials, group))
    if mode == "accumulate":
        adv = _c(advantages)
        stats = moments(adv, group) if norm_advantages else None
        out = torch.empty(N, P, dtype=adv.dtype, device=adv.device)
        call("tce_segment_accumulate_" + s, ptr(adv), ptr(pairs), P, ptr(out),
             N, T, ptr(stats), 1e-8, float(clip_advantages), stream())
        if norm_advantages:
            out = normalize(out, moments(out, group))
        return out
    if mode == "accumulated_rewards":
        r = _c(rewards)
        out = torch.empty(N, P, dtype=r.dtype, device=r.device)
        from .dist import active
        if not active(group):
            call("tce_segment_accrew_" + s, ptr(r), ptr(pairs), P, ptr(out),
                 N, T, float(gamma), None, 1, stream())
            return out
        # env shards: the reference subtracts the column mean of the WHOLE
        # batch (temporal_correlated_agent.py:311) -> raw sums, all-reduced
        # column sums + row count, second pass with the global means
        from .dist import all_reduce
        call("tce_segment_accrew_" + s, ptr(r), ptr(pairs), P, ptr(out), N, T,
             float(gamma), None, 0, stream())
        tot = torch.cat([sum_dim0(out).double(),
                         torch.full((1,), float(N), dtype=torch.float64,
                                    device=r.device)])
        all_reduce(tot, group=group)
        mean = _c((tot[:P] / tot[P]).to(r.dtype))
        call("tce_segment_accrew_" + s, ptr(r), ptr(pairs), P, ptr(out), N, T,
             float(gamma), ptr(mean), 1, stream())
        return out
    raise NotImplementedError(mode)


# ---------------------------------------------------------------------------
# shared (non-contextual) Cholesky factors
# ---------------------------------------------------------------------------
def expand_shared(L_base, N):
    """[K,K] / [1,K,K] -> stride-0 [N,K,K] view that remembers its base, so the
    ops run on the single matrix and autograd never materialises N copies."""
    base = L_base.reshape(L_base.shape[-2], L_base.shape[-1])
    out = base.unsqueeze(0).expand(N, -1, -1)
    out._tce_base = base
    return out


def split_L(L):
    """-> (L2d_or_3d contiguous tensor carrying the autograd graph, stride).
    stride 0: one matrix shared by all envs."""
    base = getattr(L, "_tce_base", None)
    if base is not None:
        return _c(base), 0
    if L.dim() == 2:
        return _c(L), 0
    if L.stride(0) == 0:
        return _c(L[0]), 0
    L = _c(L)
    return L, L.shape[-1] * L.shape[-2]


def sum_dim0(x):
    """[N, M] -> [M] column sums (two-stage, whole-chip)."""
    check_dev(x)
    x = _c(x)
    N, M = x.shape
    out = torch.empty(M, dtype=x.dtype, device=x.device)
    slices = _lib.load().tce_sum_dim0_slices(N, M)
    ws = torch.empty(slices, M, dtype=x.dtype, device=x.device) \
        if slices > 1 else None
    call("tce_sum_dim0_" + sfx(x.dtype), ptr(x), ptr(out), ptr(ws), N, M,
         stream())
    return out


# ---------------------------------------------------------------------------
# time grid, parameter sampling, ProDMP trajectories
# ---------------------------------------------------------------------------
def times(init_time, dt, num_times):
    """TemporalCorrelatedSampler.get_times: [N, T], bit-identical to the
    reference's tensor_linspace formula.  The result is tagged as the affine
    grid the kernels may share a basis table for."""
    check_dev(init_time)
    t0 = _c(init_time)
    N = t0.shape[0]
    out = torch.empty(N, num_times, dtype=t0.dtype, device=t0.device)
    # Python scalars exactly as the reference forms them (dt, num_times * dt)
    call("tce_times_" + sfx(t0.dtype), ptr(t0), float(dt),
         float(num_times * dt), ptr(out), N, int(num_times), stream())
    out._tce_affine = True
    return out


def mvn_rsample(mean, L, eps):
    """loc + scale_tril @ eps (the noise of MultivariateNormal.rsample is
    passed in explicitly)."""
    check_dev(mean, L, eps)
    mean, eps = _c(mean), _c(eps)
    Lc, sL = split_L(L)
    N, K = mean.shape
    out = torch.empty_like(mean)
    call("tce_mvn_rsample_" + sfx(mean.dtype), ptr(mean), ptr(Lc), sL,
         ptr(eps), ptr(out), N, K, stream())
    return out


def _mp_ws(mp, T, device):
    ws = getattr(mp, "_ws", None)
    if ws is None or ws[0].shape[0] < T or ws[0].device != device:
        ws = (torch.empty(T, 4 + 2 * mp.num_basis_g, dtype=mp.dtype,
                          device=device),
              torch.zeros(1, dtype=torch.int32, device=device))
        mp._ws = ws
        mp._ws_key = None               # a fresh table: nothing cached
    return ws


def _times_flags(mp, times_, init_time):
    """bit 0: general (non-affine) time rows; bit 1: the basis table in the
    mp's workspace was built for exactly these two tensors (same objects, not
    modified since), so the kernels need not rebuild it."""
    import weakref
    general = 0 if getattr(times_, "_tce_affine", False) else 1
    key = getattr(mp, "_ws_key", None)
    sig = (times_._version, init_time._version, times_.shape)
    ready = key is not None and key[0]() is times_ and key[1]() is init_time \
        and key[2] == sig and getattr(mp, "_ws", None) is not None \
        and mp._ws[0].shape[0] >= times_.shape[1]
    mp._ws_key = (weakref.ref(times_), weakref.ref(init_time), sig)
    return general | (2 if ready else 0)


def prodmp_traj(mp, times_, params, init_time, init_pos, init_vel):
    """cat[pos, vel] [N, T, 2*dof] of the ProDMP with parameters [N, K]."""
    check_dev(times_, params, init_time, init_pos, init_vel)
    t, p = _c(times_), _c(mp.pad_params(params))
    t0, y0, v0 = _c(init_time), _c(init_pos), _c(init_vel)
    N, T = t.shape
    assert p.shape == (N, mp.num_dof * mp.num_basis_g)
    out = torch.empty(N, T, 2 * mp.num_dof, dtype=p.dtype, device=p.device)
    B, flag = _mp_ws(mp, T, p.device)
    general = _times_flags(mp, t, t0)
    call("tce_prodmp_traj_" + sfx(p.dtype), *mp.c_args(), ptr(t), general,
         ptr(p), ptr(t0), ptr(y0), ptr(v0), ptr(out), ptr(B), ptr(flag), N, T,
         mp.num_dof, stream())
    return out


def _pl_work(ref, N, P, mp, sL, bwd):
    n = _lib.load().tce_pair_logprob_work_len(N, P, mp.num_dof,
                                               mp.num_basis_g, sL, int(bwd))
    if n == 0:
        return None
    return torch.empty(n, dtype=ref.dtype, device=ref.device)


class _PairLogProb(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mean, Lc, sL, mp, traj, times_, general, t0, y0, v0,
                pairs):
        N, T = times_.shape
        P = pairs.shape[0]
        logp = torch.empty(N, P, dtype=mean.dtype, device=mean.device)
        B, flag = _mp_ws(mp, T, mean.device)
        work = _pl_work(mean, N, P, mp, sL, False)
        general = (general & 1) | (_times_flags(mp, times_, t0) & 2)
        call("tce_pair_logprob_fwd_" + sfx(mean.dtype), ptr(traj), ptr(mean),
             ptr(Lc), sL, ptr(pairs), *mp.c_args(), ptr(times_), general,
             ptr(t0), ptr(y0), ptr(v0), mp.cov_reg, ptr(logp), ptr(B),
             ptr(flag), ptr(work), N, T, P, mp.num_dof, stream())
        ctx.save_for_backward(mean, Lc, traj, times_, t0, y0, v0, pairs)
        ctx.mp, ctx.sL, ctx.general = mp, sL, general
        return logp

    @staticmethod
    def backward(ctx, g):
        mean, Lc, traj, times_, t0, y0, v0, pairs = ctx.saved_tensors
        mp, sL = ctx.mp, ctx.sL
        N, T = times_.shape
        P, K = pairs.shape[0], mean.shape[1]
        g = _c(g)
        gmean = torch.empty_like(mean)
        gL = torch.empty((K, K) if sL == 0 else (N, K, K), dtype=mean.dtype,
                         device=mean.device)
        B, flag = _mp_ws(mp, T, mean.device)
        work = _pl_work(mean, N, P, mp, sL, True)
        general = (ctx.general & 1) | (_times_flags(mp, times_, t0) & 2)
        call("tce_pair_logprob_bwd_" + sfx(mean.dtype), ptr(traj), ptr(mean),
             ptr(Lc), sL, ptr(pairs), *mp.c_args(), ptr(times_), general,
             ptr(t0), ptr(y0), ptr(v0), mp.cov_reg, ptr(g), ptr(gmean),
             ptr(gL), ptr(B), ptr(flag), ptr(work), N, T, P, mp.num_dof,
             stream())
        return (gmean, gL) + (None,) * 9


def pair_log_prob(mp, traj, mean, L, times_, init_time, init_pos, init_vel,
                  pred_pairs):
    """TemporalCorrelatedPolicy.log_prob -> [N, P]; differentiable w.r.t. mean
    and L (hand-written backward kernel)."""
    check_dev(traj, mean, L, times_, init_time, init_pos, init_vel, pred_pairs)
    assert not (mp.disable_goal or mp.disable_weights), \
        "pair log-prob needs the full [weights, goal] parameterisation"
    Lc, sL = split_L(L)
    times_, init_time = _c(times_), _c(init_time)
    general = 0 if getattr(times_, "_tce_affine", False) else 1
    return _PairLogProb.apply(_c(mean), Lc, sL, mp, _c(traj), times_,
                              general, init_time, _c(init_pos),
                              _c(init_vel), _c(pred_pairs.to(torch.int64)))


# ---------------------------------------------------------------------------
# Gaussian head, param-space Gaussian, KL trust region (csrc/gauss.hip)
# ---------------------------------------------------------------------------
def first_matrix(L):
    """[K,K] matrix of a shared (non-contextual) factor, or L[0]."""
    base = getattr(L, "_tce_base", None)
    if base is not None:
        return base
    return L if L.dim() == 2 else L[0]


def full_L(L, N):
    return _c(L if L.dim() == 3 else L.unsqueeze(0).expand(N, -1, -1))


def detach_L(L):
    base = getattr(L, "_tce_base", None)
    if base is not None:
        return expand_shared(base.detach(), L.shape[0])
    return L.detach()


class _CholBuild(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vec, K, min_std):
        B, nvec = vec.shape
        L = torch.empty(B, K, K, dtype=vec.dtype, device=vec.device)
        call("tce_chol_build_fwd_" + sfx(vec.dtype), ptr(vec), ptr(L), B, K,
             nvec, float(min_std), stream())
        ctx.save_for_backward(vec)
        ctx.K = K
        return L

    @staticmethod
    def backward(ctx, gL):
        vec, = ctx.saved_tensors
        B, nvec = vec.shape
        gvec = torch.empty_like(vec)
        call("tce_chol_build_bwd_" + sfx(vec.dtype), ptr(vec), ptr(_c(gL)),
             ptr(gvec), B, ctx.K, nvec, stream())
        return gvec, None, None


def chol_build(vec, dim_out, min_std):
    """_vector_to_cholesky: [..., K | K + K(K-1)/2] -> [..., K, K]."""
    check_dev(vec)
    lead = vec.shape[:-1]
    L = _CholBuild.apply(_c(vec.reshape(-1, vec.shape[-1])), dim_out, min_std)
    return L.reshape(*lead, dim_out, dim_out)


def _vec_env(mode, bwd, x, y, Lc, sL, eps, gout, want_gL=False):
    N, K = x.shape
    s = sfx(x.dtype)
    if not bwd:
        out = torch.empty((N, K) if mode == 1 else (N,), dtype=x.dtype,
                          device=x.device)
        call("tce_vec_env_" + s, mode, 0, ptr(x), ptr(y), ptr(Lc), sL,
             float(eps), None, ptr(out), None, None, N, K, stream())
        return out
    gx = torch.empty_like(x)
    gL = torch.empty(N, K, K, dtype=x.dtype, device=x.device) if want_gL \
        else None
    call("tce_vec_env_" + s, mode, 1, ptr(x), ptr(y), ptr(Lc), sL, float(eps),
         ptr(_c(gout)), None, ptr(gx), ptr(gL), N, K, stream())
    return gx, gL


class _Maha(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, Lc, sL):
        ctx.save_for_backward(x, y, Lc)
        ctx.sL = sL
        return _vec_env(0, False, x, y, Lc, sL, 0.0, None)

    @staticmethod
    def backward(ctx, g):
        x, y, Lc = ctx.saved_tensors
        gx, _ = _vec_env(0, True, x, y, Lc, ctx.sL, 0.0, g)
        return gx, -gx, None, None


def maha(x, y, L):
    """|L^-1 (x - y)|^2 [N]; differentiable w.r.t. x and y (L is a constant:
    every call site passes the old / detached factor)."""
    check_dev(x, y, L)
    Lc, sL = split_L(L)
    return _Maha.apply(_c(x), _c(y), Lc.detach(), sL)


class _MeanProj(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mean, mean_o, Lc, sL, eps):
        ctx.save_for_backward(mean, mean_o, Lc)
        ctx.sL, ctx.eps = sL, eps
        return _vec_env(1, False, mean, mean_o, Lc, sL, eps, None)

    @staticmethod
    def backward(ctx, g):
        mean, mean_o, Lc = ctx.saved_tensors
        gx, _ = _vec_env(1, True, mean, mean_o, Lc, ctx.sL, ctx.eps, g)
        return gx, None, None, None, None


def kl_mean_projection(mean, mean_old, L_old, eps):
    check_dev(mean, mean_old, L_old)
    Lc, sL = split_L(L_old)
    return _MeanProj.apply(_c(mean), _c(mean_old).detach(), Lc.detach(), sL,
                           float(eps))


class _MvnLogProb(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mean, Lc, sL):
        ctx.save_for_backward(x, mean, Lc)
        ctx.sL = sL
        return _vec_env(2, False, x, mean, Lc, sL, 0.0, None)

    @staticmethod
    def backward(ctx, g):
        x, mean, Lc = ctx.saved_tensors
        N, K = x.shape
        need_L = ctx.needs_input_grad[2]
        gmean, gL = _vec_env(2, True, x, mean, Lc, ctx.sL, 0.0, g,
                             want_gL=need_L)
        if need_L and ctx.sL == 0:
            gL = sum_dim0(gL.reshape(N, K * K)).reshape(K, K)
        return None, gmean, gL, None


def mvn_log_prob(x, mean, L):
    """MultivariateNormal(mean, scale_tril=L).log_prob(x) [N]."""
    check_dev(x, mean, L)
    Lc, sL = split_L(L)
    return _MvnLogProb.apply(_c(x).detach(), _c(mean), Lc, sL)


def log_determinant(L):
    return 2 * L.diagonal(dim1=-2, dim2=-1).log().sum(-1)


def mvn_entropy(L, N=None):
    """MultivariateNormal.entropy(): K/2 (1 + log 2 pi) + sum log diag L."""
    import math
    base = getattr(L, "_tce_base", None)
    M = base if base is not None else L
    K = M.shape[-1]
    ent = 0.5 * K * (1.0 + math.log(2 * math.pi)) + \
        M.diagonal(dim1=-2, dim2=-1).log().sum(-1)
    if base is not None:
        return ent.expand(L.shape[0])
    return ent


class _KLCovPart(torch.autograd.Function):
    @staticmethod
    def forward(ctx, L, Lo, sLo):
        B, K = L.shape[0], L.shape[-1]
        out = torch.empty(B, dtype=L.dtype, device=L.device)
        call("tce_kl_cov_part_" + sfx(L.dtype), 0, ptr(L), ptr(Lo), sLo, None,
             ptr(out), None, B, K, stream())
        ctx.save_for_backward(L, Lo)
        ctx.sLo = sLo
        return out

    @staticmethod
    def backward(ctx, g):
        L, Lo = ctx.saved_tensors
        B, K = L.shape[0], L.shape[-1]
        gL = torch.empty_like(L)
        call("tce_kl_cov_part_" + sfx(L.dtype), 1, ptr(L), ptr(Lo), ctx.sLo,
             ptr(_c(g)), None, ptr(gL), B, K, stream())
        return gL, None, None


def kl_cov_part(L, L_old, N):
    """Covariance part of KL(p || q) [N] (constant across envs when both
    factors are shared); differentiable w.r.t. L."""
    check_dev(L, L_old)
    Lc, sL = split_L(L)
    Loc, sLo = split_L(L_old)
    K = Lc.shape[-1]
    if sL == 0:
        if sLo != 0:
            Loc, sLo = _c(Loc[0]), 0
        out = _KLCovPart.apply(Lc.reshape(1, K, K), Loc.detach(), 0)
        return out.expand(N)
    return _KLCovPart.apply(Lc, Loc.detach(), sLo)


class _KLCovProj(torch.autograd.Function):
    @staticmethod
    def forward(ctx, L, Lo, sLo, eps_cov, beta, entropy_eq):
        B, K = L.shape[0], L.shape[-1]
        proj = torch.empty_like(L)
        n = _lib.load().tce_kl_cov_proj_ctx_len(K)
        cbuf = torch.empty(B, n, dtype=torch.float64, device=L.device)
        call("tce_kl_cov_proj_fwd_" + sfx(L.dtype), ptr(L), ptr(Lo), sLo,
             float(eps_cov), ptr(beta), int(bool(entropy_eq)), ptr(proj),
             ptr(cbuf), B, K, 0, stream())
        ctx.save_for_backward(L, Lo, proj, cbuf)
        ctx.sLo = sLo
        return proj

    @staticmethod
    def backward(ctx, g):
        L, Lo, proj, cbuf = ctx.saved_tensors
        B, K = L.shape[0], L.shape[-1]
        gL = torch.empty_like(L)
        call("tce_kl_cov_proj_bwd_" + sfx(L.dtype), ptr(L), ptr(Lo), ctx.sLo,
             ptr(proj), ptr(cbuf), ptr(_c(g)), ptr(gL), B, K, stream())
        return gL, None, None, None, None, None


def kl_cov_projection(L, L_old, eps_cov, beta=None, entropy_eq=False):
    """[B,K,K] -> projected Cholesky factors [B,K,K] (KL bound eps_cov on the
    covariance part, then entropy control with the device scalar beta);
    differentiable w.r.t. L."""
    check_dev(L, L_old, beta)
    L = _c(L)
    Lo = _c(L_old).detach()
    sLo = 0 if Lo.dim() == 2 else Lo.shape[-1] * Lo.shape[-2]
    if beta is not None:
        beta = _c(beta.detach().to(L.dtype).reshape(1))
    return _KLCovProj.apply(L, Lo, sLo, eps_cov, beta, entropy_eq)


# ---------------------------------------------------------------------------
# rollout buffer: running mean/std of observations, MDP reward (rollout.hip)
# ---------------------------------------------------------------------------
def rms_update(x, mean, var, count):
    """RunningMeanStd.update of x [R, D] into mean/var [D] (in place); returns
    the new count (host float, as in the reference)."""
    check_dev(x, mean, var)
    x = _c(x)
    R, D = x.shape
    ws = torch.empty(_lib.load().tce_rms_num_partials(), D, 2,
                     dtype=torch.float64, device=x.device)
    call("tce_rms_update_" + sfx(x.dtype), ptr(x), R, D, ptr(mean), ptr(var),
         float(count), ptr(ws), stream())
    return count + R


ENV_FAMILIES = {"reach": 0, "push": 1, "table_tennis": 2, "hopper": 3}


def env_rollout(actions, init_obs, family, dof, d_task, dt, kp, kd,
                want_states=True, want_flags=False, shift=None,
                want_moments=False):
    """One whole episode of the synthetic env suite (csrc/env.hip) for the N
    envs of actions [N, T, 2 dof] / init_obs [N, D] -> dict with
    ``states`` [N, T+1, D] (row 0 = init_obs) or None, ``rewards`` [N, T],
    ``flags`` [N, T] bool or None, ``metrics`` [N, 2] {success, final
    distance}, ``partials`` float64 [N, D, 2] (column moments of the T+1 rows,
    shifted by ``shift``) or None."""
    check_dev(actions, init_obs, shift)
    actions, init_obs = _c(actions), _c(init_obs)
    N, T, A = actions.shape
    D = init_obs.shape[-1]
    assert A == 2 * dof and D == d_task + 1 + 2 * dof and init_obs.shape[0] == N
    dt_, dev = actions.dtype, actions.device
    states = torch.empty(N, T + 1, D, dtype=dt_, device=dev) \
        if want_states else None
    rewards = torch.empty(N, T, dtype=dt_, device=dev)
    flags = torch.empty(N, T, dtype=torch.uint8, device=dev) \
        if want_flags else None
    metrics = torch.empty(N, 2, dtype=dt_, device=dev)
    partials = torch.empty(N, D, 2, dtype=torch.float64, device=dev) \
        if want_moments else None
    if shift is not None:
        shift = _c(shift.to(dt_))
    call("tce_env_rollout_" + sfx(dt_), ptr(actions), ptr(init_obs),
         ENV_FAMILIES[family], N, T, int(dof), int(d_task), float(dt),
         float(kp), float(kd), ptr(states), ptr(rewards), ptr(flags),
         ptr(metrics), ptr(shift), ptr(partials), stream())
    return dict(states=states, rewards=rewards,
                flags=None if flags is None else flags.view(torch.bool),
                metrics=metrics, partials=partials)


def rms_merge(partials, rows, mean, var, count, shift=None):
    """Fold the moment partials [nparts, D, 2] of `rows` observation rows
    (accumulated relative to `shift`, default: the running mean itself) into
    the running mean / var [D] in place; returns the new count."""
    check_dev(partials, mean, var, shift)
    D = mean.numel()
    assert partials.dtype == torch.float64 and partials.shape[-2:] == (D, 2)
    partials = _c(partials)
    sh = mean if shift is None else _c(shift)
    call("tce_rms_merge_" + sfx(mean.dtype), ptr(partials),
         partials.numel() // (2 * D), D, ptr(sh), float(rows), float(count),
         ptr(mean), ptr(var), stream())
    return count + rows


def rms_normalize(x, mean, var, eps=1e-8, inplace=False):
    check_dev(x, mean, var)
    x = _c(x)
    y = x if inplace else torch.empty_like(x)
    call("tce_rms_normalize_" + sfx(x.dtype), ptr(x), ptr(y), x.numel(),
         x.shape[-1], ptr(mean), ptr(var), float(eps), stream())
    return y


_MEDIAN_WS = {}


def median(x):
    """Lower median of all elements of a float32 / float64 device tensor as a
    0-dim float64 tensor (csrc/select.hip: radix select, no sort)."""
    x = _c(x.detach()).reshape(-1)
    s = stream()
    key = (x.device, s)
    ws = _MEDIAN_WS.get(key)
    if ws is None:
        ws = _MEDIAN_WS[key] = torch.zeros(
            _lib.load().tce_median_ws_len(), dtype=torch.int32, device=x.device)
    out = torch.empty((), dtype=torch.float64, device=x.device)
    call("tce_median_" + sfx(x.dtype), ptr(x), x.numel(), ptr(out), ptr(ws), s)
    return out


_STATS5_WS = {}


def stats5(x, out=None):
    """{mean, max, min, median, std (n - 1)} of all elements of a device tensor
    as a float64 [5] tensor (csrc/select.hip: the radix select's first pass
    carries the sums and extrema) -- the five entries of generate_stats
    (util_numerical.py:130-164) in ONE chain of launches.  float32 / float64
    as they are; bool / integer tensors as float64 (step counts and lengths
    summed over envs exceed float32's 2^24)."""
    x = x.detach()
    if x.dtype not in (torch.float32, torch.float64):
        x = x.to(torch.float64)
    x = _c(x).reshape(-1)
    s = stream()
    key = (x.device, s)
    ws = _STATS5_WS.get(key)
    if ws is None:
        ws = _STATS5_WS[key] = torch.zeros(
            (_lib.load().tce_stats5_ws_len() + 1) // 2, dtype=torch.int64,
            device=x.device)
    if out is None:
        out = torch.empty(5, dtype=torch.float64, device=x.device)
    call("tce_stats5_" + sfx(x.dtype), ptr(x), x.numel(), ptr(out), ptr(ws), s)
    return out


def mdp_reward(step_rewards, event_flags):
    """make_mdp_reward: returns the re-shaped rewards (new tensor)."""
    check_dev(step_rewards, event_flags)
    if tuple(event_flags.shape) != tuple(step_rewards.shape) or \
            step_rewards.dim() != 2:
        raise ValueError(
            "mdp_reward: per-step event flags %s must match step_rewards %s "
            "([N, T])" % (tuple(event_flags.shape), tuple(step_rewards.shape)))
    r = _c(step_rewards).clone()
    N, T = r.shape
    call("tce_mdp_reward_" + sfx(r.dtype), ptr(r), ptr(_bool_u8(event_flags)),
         N, T, stream())
    return r


# ---------------------------------------------------------------------------
# MLP forward (csrc/mlp.hip: fused MFMA kernels; see mlp_forward below)
# ---------------------------------------------------------------------------
_ACT = {"tanh": 0, "relu": 1, "leaky_relu": 2, "softplus": 3, None: -1}


def mlp_forward(mlp, x):
    """MLP.forward (mprl/util/util_nn.py:225-246): hidden layers with the
    hidden activation, then the output layer (+ optional last activation)."""
    from . import mlp_ops
    return mlp_ops.forward(mlp, x)
