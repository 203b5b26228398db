"""Policy mean nets on the row kernels of csrc/pmlp.hip: D_in <= 64 -> H (-> H)
-> K <= 64 in float32 / float64 (H 128 with one or two hidden layers, H 256 with
one) -- the mean nets of the box-pushing (float64, 128 x 2) and table-tennis
(256 x 1 tanh) policies (mprl/config/box_push_random_init/tcp/entire/shared.yaml:
7,75-78, mprl/config/table_tennis_4d/tcp/entire/shared.yaml:78-81), which the
fused 128 x 2 float32 kernels of csrc/mlp.hip do not cover.  MLP.forward
(mprl/util/util_nn.py:225-246) and its backward without autograd and without a
library GEMM.
"""
import torch

from . import _lib
from ._lib import call, ptr, sfx, stream

_ACT = {"tanh": 0, "relu": 1, "leaky_relu": 2, "softplus": 3}


def shape(mlp):
    """(din, H, num_hidden, dout) of an MLP the kernels cover, else None."""
    hl = list(mlp.hidden_layers)
    if not (1 <= len(hl) <= 2 and all(h == hl[0] for h in hl)):
        return None
    if mlp.act_func_hidden_type not in _ACT or mlp.act_func_last_type is not None:
        return None
    if mlp.dtype not in (torch.float32, torch.float64):
        return None
    s = (mlp.dim_in, hl[0], len(hl), mlp.dim_out)
    esz = 4 if mlp.dtype == torch.float32 else 8
    return s if _lib.load().tce_pmlp_supported(*s, esz) else None


def supported(mlp):
    return shape(mlp) is not None


def flat_params(mlp):
    """The parameters as ONE buffer in MLP.parameters() order: the optimizer's
    flat buffer when the parameters are its views (no copy), else a copy."""
    ps = list(mlp.parameters())
    base, esz, off = ps[0].data_ptr(), ps[0].element_size(), 0
    store = ps[0].untyped_storage().data_ptr()
    flat_ok = base % 16 == 0
    for p in ps:
        if p.data_ptr() != base + off * esz or not p.is_contiguous() or \
                p.untyped_storage().data_ptr() != store:
            flat_ok = False
        off += p.numel()
    if not flat_ok:
        return torch.cat([q.detach().reshape(-1) for q in ps])
    return torch.empty(0, dtype=ps[0].dtype, device=ps[0].device).set_(
        ps[0].untyped_storage(), ps[0].storage_offset(), (off,), (1,))


def _rows(x, din):
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    assert x2.shape[-1] >= din
    return x2


def forward(mlp, x, keep=None, param=None):
    """out [.., K] = MLP(x).  keep: dict that receives the hidden activations
    (h1, h2) for ``backward``."""
    din, H, NL, K = shape(mlp)
    x2 = _rows(x, din)
    N = x2.shape[0]
    dt, dev = mlp.dtype, x.device
    if x2.dtype != dt or not x2.is_cuda:
        raise RuntimeError("pmlp: input must be a %s HIP tensor" % dt)
    param = flat_params(mlp) if param is None else param
    out = torch.empty(N, K, dtype=dt, device=dev)
    h1 = h2 = None
    if keep is not None:
        h1 = torch.empty(N, H, dtype=dt, device=dev)
        h2 = torch.empty(N, H, dtype=dt, device=dev) if NL == 2 else None
        keep.update(h1=h1, h2=h2, x=x2)
    call("tce_pmlp_forward_" + sfx(dt), ptr(x2), x2.stride(0), N, din, H, NL, K,
         _ACT[mlp.act_func_hidden_type], ptr(param), ptr(h1), ptr(h2), ptr(out),
         stream())
    return out.reshape(*x.shape[:-1], K)


def backward(mlp, keep, grad_out, grad=None, partials=None, param=None):
    """Gradient of sum(grad_out * out) w.r.t. the flat parameters -> grad [P]."""
    din, H, NL, K = shape(mlp)
    x2 = keep["x"]
    N = x2.shape[0]
    dt, dev = mlp.dtype, x2.device
    lib = _lib.load()
    P = lib.tce_pmlp_num_params(din, H, NL, K)
    param = flat_params(mlp) if param is None else param
    g = grad_out.reshape(N, K)
    g = g if g.is_contiguous() else g.contiguous()
    if grad is None:
        grad = torch.empty(P, dtype=dt, device=dev)
    if partials is None:
        partials = torch.empty(lib.tce_pmlp_max_slabs() * P, dtype=dt, device=dev)
    call("tce_pmlp_backward_" + sfx(dt), ptr(x2), x2.stride(0), N, din, H, NL, K,
         _ACT[mlp.act_func_hidden_type], ptr(param), ptr(keep["h1"]),
         ptr(keep["h2"]), ptr(g), ptr(partials), ptr(grad), stream())
    return grad


# ---------------------------------------------------------------------------
# the black-box agent's value function on these kernels (csrc/vcritic.hip)
# ---------------------------------------------------------------------------
def critic_supported(agent):
    """A value function D_in -> H (-> H) -> 1 of a shape above with its
    parameters in a FlatAdam, full-batch epochs (table tennis's 256 x 1 BBRL
    critic, mprl/config/table_tennis_4d/bbrl/entire/shared.yaml:90-91)."""
    from .smlp_ops import _opt_matches
    net, opt = agent.critic.net, agent.critic_optimizer
    from .smlp_ops import minibatches_ok
    return (net.dim_out == 1 and supported(net) and minibatches_ok(agent)
            and _opt_matches(opt, list(net.parameters()))
            and opt.flat_param.numel() <= (1 << 17)
            and opt.flat_param.data_ptr() % 16 == 0)


def critic_update(agent, states, returns, old_values):
    """E critic epochs (black_box_agent.py:105-157) -> rec [E, 3] = {loss,
    |g|, |g| clipped} on the device; one C call per epoch
    (tce_pmlp_critic_epoch_*), no autograd, no library GEMM."""
    net, opt = agent.critic.net, agent.critic_optimizer
    din, H, NL, _ = shape(net)
    lib = _lib.load()
    x = _rows(states, din)
    N, E = x.shape[0], agent.epochs_critic
    dt, dev = net.dtype, x.device
    if x.dtype != dt:
        raise RuntimeError("pmlp critic: states must be %s" % dt)
    ret = returns.reshape(-1).contiguous()
    old = old_values.reshape(-1).contiguous() if agent.clip_critic > 0 \
        else None
    cache = net.__dict__.setdefault("_tce_pmlp_ws", {})

    def workspace(n):
        ws = cache.get(("ws", n))
        if ws is None:
            # zeroed once: the loss kernel re-arms its ticket itself (a few
            # batch sizes at a time -- the two piece lengths of a minibatched
            # epoch --, older ones are let go)
            old_keys = [k for k in cache if k != "partials"]
            if len(old_keys) > 2:
                for k in old_keys:
                    del cache[k]
            ws = cache[("ws", n)] = torch.zeros(
                lib.tce_pmlp_critic_ws_len(n, H), dtype=dt, device=dev)
        return ws
    partials = cache.get("partials")
    if partials is None:
        P = lib.tce_pmlp_num_params(din, H, NL, 1)
        partials = cache["partials"] = torch.empty(
            lib.tce_pmlp_max_slabs() * P, dtype=dt, device=dev)
    kmb = int(agent.num_minibatchs)
    rec = torch.zeros(E * kmb, 3, dtype=dt, device=dev)
    g = opt.param_groups[0]
    opt.bind_grads()
    # env shards: the Adam launch adds the peers' gradients (agent.xchg_critic);
    # without an in-library exchange the call stops in front of the step
    xch = agent.xchg_critic if agent.dist.active else None
    sharded = agent.dist.active and xch is None
    gscale = 1.0 / agent.dist.world if xch is not None else 1.0

    def one(xr, rr, orr, rec_row):
        n = xr.shape[0]
        call("tce_pmlp_critic_epoch_" + sfx(dt), ptr(xr), xr.stride(0),
             ptr(rr), ptr(orr), n, din, H, NL,
             _ACT[net.act_func_hidden_type], float(agent.clip_critic),
             ptr(opt.flat_param), ptr(opt.flat_grad), ptr(opt.m), ptr(opt.v),
             ptr(opt.dev_state), float(g["lr"]), float(g["betas"][0]),
             float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
             float(agent.clip_grad_norm), gscale, int(not sharded),
             float(opt.host_step), ptr(workspace(n)), ptr(partials),
             ptr(rec_row), None if xch is None else xch.handle, stream())
    if kmb > 1:
        # one optimizer step per piece of the epoch's permutation
        # (black_box_agent.py:124-146) on a gathered copy of its rows
        from .smlp_ops import gather_rows, minibatch_pieces
        assert not sharded
        row = 0
        for _ in range(E):
            for idx in minibatch_pieces(N, kmb, dev):
                opt.host_step += 1
                one(*gather_rows(net, x[:, :din], ret, old, idx), rec[row])
                row += 1
        opt._opt_called = True
        return rec
    for e in range(E):
        if not sharded:
            opt.host_step += 1
            opt._opt_called = True            # for LinearLR's order check
        one(x, ret, old, rec[e])
        if sharded:
            # the loss is the shard's own mean (as on the other sharded paths);
            # sum of the shards' gradients, then clip + Adam + the two norms
            agent.dist.allreduce_flat(opt.flat_grad, average=False)
            opt.step_once(agent.clip_grad_norm,
                          grad_scale=1.0 / agent.dist.world,
                          norms_out=rec[e, 1:3])
    return rec
