"""Small two-hidden-layer networks of the black-box agent on the hand-written
row kernels of csrc/smlp.hip (D_in <= 64 -> H -> H -> D_out, H in {32, 64},
float32): rollout forward, the critic update and the policy update of
``BlackBoxAgent`` (mprl/rl/agent/black_box_agent.py:105-389) without autograd
and without library GEMMs -- two launches per critic epoch, three (diagonal
covariance, the reference's BBRL configuration) or seven per policy epoch.
"""
import torch

from . import _lib
from ._lib import call, ptr, stream

_ACT = {"tanh": 0, "relu": 1, "leaky_relu": 2, "softplus": 3}
HEAD_NONE, HEAD_VALUE, HEAD_BB_POLICY = 0, 1, 2


def supported(mlp, head=HEAD_NONE):
    hl = list(mlp.hidden_layers)
    if not (mlp.dtype == torch.float32 and len(hl) == 2 and hl[0] == hl[1]
            and mlp.act_func_last_type is None
            and mlp.act_func_hidden_type in _ACT):
        return False
    if head == HEAD_VALUE and mlp.dim_out != 1:
        return False
    return bool(_lib.load().tce_smlp_supported(mlp.dim_in, hl[0], mlp.dim_out,
                                               head))


def flat_params(mlp):
    """The parameters as ONE flat buffer in the order W1 | b1 | W2 | b2 | W3 |
    b3: the tensors themselves when they already lie back to back (views of a
    FlatAdam buffer), a copy otherwise."""
    ps = list(mlp.parameters())
    ok = all(p.is_contiguous() for p in ps)
    if ok:
        for a, b in zip(ps[:-1], ps[1:]):
            if b.data_ptr() != a.data_ptr() + a.numel() * a.element_size():
                ok = False
                break
    if ok:
        n = sum(p.numel() for p in ps)
        return ps[0].detach().as_strided((n,), (1,))
    return torch.cat([p.detach().reshape(-1) for p in ps])


def _rows2d(x, din):
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    assert x2.shape[1] >= din
    return x2


def forward(mlp, x):
    """MLP.forward without autograd (rollouts, evaluation)."""
    x2 = _rows2d(x, mlp.dim_in)
    N = x2.shape[0]
    out = torch.empty(N, mlp.dim_out, dtype=torch.float32, device=x.device)
    call("tce_smlp_forward_f32", ptr(x2), x2.stride(0), N, mlp.dim_in,
         mlp.hidden_layers[0], mlp.dim_out, _ACT[mlp.act_func_hidden_type],
         ptr(flat_params(mlp)), ptr(out), stream())
    return out.reshape(*x.shape[:-1], mlp.dim_out)


def _opt_matches(opt, params):
    from .optim import FlatAdam
    return isinstance(opt, FlatAdam) and len(opt._params) == len(params) and \
        all(a is b for a, b in zip(opt._params, params))


def _workspace(owner, key, n, dtype=torch.float32):
    """A zero-initialised scratch buffer kept on `owner` (the kernels re-arm
    their ticket themselves, so it is zeroed once)."""
    cache = owner.__dict__.setdefault("_tce_smlp_ws", {})
    buf = cache.get(key)
    if buf is None or buf.numel() < n:
        dev = next(owner.parameters()).device
        buf = torch.zeros(n, dtype=dtype, device=dev)
        cache[key] = buf
    return buf


def minibatches_ok(agent):
    """The row kernels take the critic's minibatches (black_box_agent.py:
    124-131; class default 10) as gathered copies of the rows, one epoch call
    per piece -- whenever the step needs no torch.distributed all-reduce
    between the kernels."""
    return agent.num_minibatchs == 1 or not agent.dist.active or \
        agent.xchg_critic is not None


def minibatch_pieces(n, k, device):
    """generate_minibatches (mprl/util/util_data_structure.py:378-391): the
    epoch's permutation from numpy's GLOBAL generator -- the reference's own
    draw, so the same pieces from the same seed -- as device index tensors."""
    import numpy as np
    idx = np.arange(n)
    np.random.shuffle(idx)
    dev = torch.as_tensor(idx, device=device)
    out, off = [], 0
    for i in range(k):
        ln = n // k + (1 if i < n % k else 0)
        out.append(dev[off:off + ln])
        off += ln
    return out


def gather_rows(owner, x, a, b, idx):
    """select_batch: (x[idx], a[idx], b[idx]) as contiguous copies in buffers
    kept on `owner` (tce_gather_rows_*: one launch, no aten indexing)."""
    from ._lib import sfx
    n, din = idx.numel(), x.shape[1]
    cache = owner.__dict__.setdefault("_tce_gather", {})
    key = (n, din, x.dtype)
    bufs = cache.get(key)
    if bufs is None:
        if len(cache) > 4:
            cache.clear()
        bufs = cache[key] = (
            torch.empty(n, din, dtype=x.dtype, device=x.device),
            torch.empty(n, dtype=x.dtype, device=x.device),
            torch.empty(n, dtype=x.dtype, device=x.device))
    xo, ao, bo = bufs
    call("tce_gather_rows_" + sfx(x.dtype), ptr(x), x.stride(0), ptr(a),
         ptr(b), ptr(idx), n, din, ptr(xo), ptr(ao),
         None if b is None else ptr(bo), stream())
    return xo, ao, (None if b is None else bo)


def critic_supported(agent):
    net = agent.critic.net
    return supported(net, HEAD_VALUE) and minibatches_ok(agent) and \
        _opt_matches(agent.critic_optimizer, list(net.parameters()))


def critic_update(agent, states, returns, old_values):
    """E critic epochs -> rec [E, 3] = {loss, |g|, |g| clipped} (device)."""
    net, opt = agent.critic.net, agent.critic_optimizer
    lib = _lib.load()
    x = _rows2d(states, net.dim_in)
    N, E = x.shape[0], agent.epochs_critic
    H = net.hidden_layers[0]
    ret = returns.reshape(-1).contiguous()
    old = old_values.reshape(-1).contiguous() if agent.clip_critic > 0 \
        else None
    k = int(agent.num_minibatchs)
    rec = torch.zeros(E * k, 3, dtype=torch.float32, device=x.device)
    g = opt.param_groups[0]
    opt.bind_grads()

    def launch(epochs, do_adam, rec_rows, scale, xch=None, rows=None):
        xr, rr, orr = (x, ret, old) if rows is None else rows
        n = xr.shape[0]
        ws = _workspace(net, ("ws", n),
                        lib.tce_smlp_ws_len(n, net.dim_in, H, 1))
        call("tce_smlp_critic_epochs_f32", ptr(xr), xr.stride(0), ptr(rr),
             ptr(orr), n, net.dim_in, H, _ACT[net.act_func_hidden_type],
             float(agent.clip_critic), ptr(opt.flat_param), ptr(opt.flat_grad),
             ptr(opt.m), ptr(opt.v), ptr(opt.dev_state), float(g["lr"]),
             float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
             float(g["weight_decay"]), float(agent.clip_grad_norm), scale,
             int(do_adam), opt.host_step + 1, epochs, ptr(ws), ptr(rec_rows),
             None if xch is None else xch.handle, stream())
    xch = agent.xchg_critic if agent.dist.active else None
    if k > 1:
        # one optimizer step per piece of the epoch's permutation
        # (black_box_agent.py:124-146), each the row kernels' one-epoch call on
        # a gathered copy of its rows
        scale = 1.0 / agent.dist.world if xch is not None else 1.0
        row = 0
        for _ in range(E):
            for idx in minibatch_pieces(N, k, x.device):
                launch(1, True, rec[row:row + 1], scale, xch,
                       rows=gather_rows(net, x[:, :net.dim_in], ret, old, idx))
                opt.host_step += 1
                row += 1
        opt._opt_called = True
        if xch is None:
            before = rec[:, 1].sqrt()
            after = before
            if agent.clip_grad_norm > 0:
                after = before * torch.clamp(
                    agent.clip_grad_norm / (before + 1e-6), max=1.0)
            rec[:, 1], rec[:, 2] = before, after
        return rec
    if not agent.dist.active or xch is not None:
        # (env shards: a small launch behind the slab reduction adds the peers'
        # gradients and applies Adam -- the E epochs stay ONE call)
        scale = 1.0 / agent.dist.world if xch is not None else 1.0
        launch(E, True, rec, scale, xch)
        opt.host_step += E
        opt._opt_called = True                # for LinearLR's order check
        if xch is None:
            # the kernels leave |g|^2: the two norms of grad_norm_clip from it
            # (env shards: the exchange's Adam launch has written both norms)
            before = rec[:, 1].sqrt()
            after = before
            if agent.clip_grad_norm > 0:
                after = before * torch.clamp(
                    agent.clip_grad_norm / (before + 1e-6), max=1.0)
            rec[:, 1], rec[:, 2] = before, after
    else:
        for e in range(E):
            launch(1, False, rec[e], 1.0)
            agent.dist.allreduce_flat(opt.flat_grad, average=False)
            # clip + Adam + the record's two norms in ONE launch (tce_adam_once_*)
            opt.step_once(agent.clip_grad_norm,
                          grad_scale=1.0 / agent.dist.world,
                          norms_out=rec[e, 1:3])
    return rec


def policy_supported(agent, L_old):
    from . import ops
    from .rl.projection import KLProjectionLayer
    pol, proj = agent.policy, agent.projection
    net = pol.mean_net
    return (supported(net, HEAD_BB_POLICY) and not pol.contextual_cov
            and type(proj) is KLProjectionLayer and not proj.entropy_first
            and ops.split_L(L_old)[1] == 0
            and _opt_matches(agent.policy_optimizer,
                             list(net.parameters()) +
                             [pol.variance_net.variable]))


def policy_update(agent, states, actions, logp_old, adv, mean_old, L_old,
                  beta, balance=False):
    """E policy epochs -> (rec [E, 19] = 7 loss / norm scalars + 12 KL means,
    mean_new, L_new [K,K], proj_mean, proj_L [K,K]) with the last epoch's
    distributions."""
    from . import ops
    pol, proj, opt = agent.policy, agent.projection, agent.policy_optimizer
    net, lib = pol.mean_net, _lib.load()
    x = _rows2d(states, net.dim_in)
    N, E, K = x.shape[0], agent.epochs_policy, pol.dim_out
    H = net.hidden_layers[0]
    dev = x.device
    c = lambda t: t if t.is_contiguous() else t.contiguous()
    actions, logp_old, adv, mean_old = c(actions), c(logp_old), c(adv), \
        c(mean_old)
    L_old = c(ops.split_L(L_old)[0].detach())
    var = pol.variance_net.variable
    ws = _workspace(net, ("ws", N), lib.tce_smlp_ws_len(N, net.dim_in, H, K))
    mats = _workspace(net, ("mats", K), lib.tce_bb_policy_mats_len(K))
    ctx = torch.zeros(lib.tce_kl_cov_proj_ctx_len(K), dtype=torch.float64,
                      device=dev)
    # per epoch: 7 loss / norm scalars + the 12 KL means of kl_old_new_proj
    rec = torch.zeros(E, 19, dtype=torch.float32, device=dev)
    mean_new = torch.empty(N, K, dtype=torch.float32, device=dev)
    proj_mean = torch.empty(N, K, dtype=torch.float32, device=dev)
    beta_t = None if beta is None else \
        c(beta.detach().to(torch.float32).reshape(1))
    include_cov = int(pol.contextual_std or not agent.set_variance)
    # diagonal factors (std_only) without an entropy bound: K-vector kernels
    # instead of the K x K ones (TCE_BB_DIAG=0: the general kernels, for tests)
    import os
    diag = int(pol.std_only and beta_t is None and
               os.environ.get("TCE_BB_DIAG", "1") != "0")
    g = opt.param_groups[0]
    opt.bind_grads()

    xch = agent.xchg_policy if agent.dist.active else None
    gscale = 1.0 / agent.dist.world if xch is not None else 1.0

    def launch(epochs, do_adam, rec_rows, adv=adv, tr_coeff=None,
               ent_coef=None, again=False):
        call("tce_bb_policy_epochs_f32", ptr(x), x.stride(0), ptr(actions),
             ptr(logp_old), ptr(adv), ptr(mean_old), ptr(L_old), N,
             net.dim_in, H, K, _ACT[net.act_func_hidden_type], var.numel(),
             float(pol.min_std), float(proj.mean_bound), float(proj.cov_bound),
             ptr(beta_t), int(bool(proj.entropy_eq)),
             float(proj.trust_region_coeff if tr_coeff is None else tr_coeff),
             include_cov,
             float(agent.entropy_penalty_coef if ent_coef is None
                   else ent_coef), ptr(opt.flat_param),
             ptr(opt.flat_grad), ptr(opt.m), ptr(opt.v), ptr(opt.dev_state),
             float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
             float(g["eps"]), float(g["weight_decay"]),
             float(agent.clip_grad_norm), gscale if do_adam else 1.0,
             int(do_adam), diag | (2 if again else 0), epochs,
             ptr(ctx), ptr(ws), ptr(mats), ptr(rec_rows), 19, ptr(mean_new),
             ptr(proj_mean),
             None if xch is None or not do_adam else xch.handle, stream())

    def shard_norm(row):
        """|mean over the ranks of the gradient the last launch left| -> row[5]
        (a balance norm of a sharded run)."""
        if xch is not None:
            xch.allreduce(opt.flat_grad)
        else:
            agent.dist.allreduce_flat(opt.flat_grad, agent._policy_group,
                                      average=False)
        row[5] = opt.flat_grad.norm() / agent.dist.world
    if balance:
        # an iteration with the policy balance check (black_box_agent.py:
        # 218-284): before every epoch the parameter-gradient norms of the
        # surrogate loss alone (no trust region / entropy term) and of the trust
        # region loss alone (zero advantages) -- the same kernels without the
        # optimizer step; the finish kernel leaves |g| in its record row
        zero_adv = torch.zeros_like(adv)
        bal = torch.zeros(2, E, 19, dtype=torch.float32, device=dev)
        for e in range(E):
            # (again: L_old^-1 stays in `mats` from the update's first call)
            launch(1, False, bal[0, e], tr_coeff=0.0, ent_coef=0.0, again=e > 0)
            if agent.dist.active:
                shard_norm(bal[0, e])
            launch(1, False, bal[1, e], adv=zero_adv, ent_coef=0.0, again=True)
            if agent.dist.active:
                shard_norm(bal[1, e])
            if not agent.dist.active or xch is not None:
                launch(1, True, rec[e], again=True)
                opt.host_step += 1
            else:
                launch(1, False, rec[e], again=True)
                agent.dist.allreduce_flat(opt.flat_grad, agent._policy_group,
                                          average=False)
                opt.step_once(agent.clip_grad_norm,
                              grad_scale=1.0 / agent.dist.world,
                              norms_out=rec[e, 5:7])
        opt._opt_called = True
        rec = torch.cat([rec, bal[0, :, 5:6], bal[1, :, 5:6]], dim=1)
    elif not agent.dist.active or xch is not None:
        # (env shards: the finish kernel adds the peers' gradients before Adam
        # -- the E epochs stay ONE call)
        launch(E, True, rec)
        opt.host_step += E
        opt._opt_called = True
    else:
        for e in range(E):
            launch(1, False, rec[e])
            agent.dist.allreduce_flat(opt.flat_grad, agent._policy_group,
                                      average=False)
            opt.step_once(agent.clip_grad_norm,
                          grad_scale=1.0 / agent.dist.world,
                          norms_out=rec[e, 5:7])
    KK = (K * K + 3) // 4 * 4
    L_new = mats[:K * K].view(K, K).clone()
    proj_L = mats[KK:KK + K * K].view(K, K).clone()
    return rec, mean_new, L_new, proj_mean, proj_L
