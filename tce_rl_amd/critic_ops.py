"""Fused critic epochs on the exact matrix cores.

``supported(mlp)``: the value network D_in -> H -> H -> 1 with
* H = 128 in float32 (the Metaworld config): csrc/mlp.hip, one launch per epoch
  (+ the split-f16 variant csrc/mlp16.hip);
* H = 256 in float32 / float64 (box pushing, table tennis) and H = 128 in
  float64: csrc/mlpw_*.hip, chain kernel + weight-gradient kernel per epoch.
Anything else stays on the library-GEMM path (mlp_ops).
"""
import torch

from . import _lib
from ._lib import call, ptr, sfx, stream

_ACT = {"tanh": 0, "relu": 1, "leaky_relu": 2, "softplus": 3}


def narrow_supported(mlp):
    """D_in -> 128 -> 128 -> 1, float32: csrc/mlp.hip."""
    return (mlp.dtype == torch.float32 and mlp.dim_out == 1
            and list(mlp.hidden_layers) == [128, 128]
            and 1 <= mlp.dim_in <= 40 and mlp.act_func_last_type is None
            and mlp.act_func_hidden_type in _ACT)


def wide_supported(mlp):
    """The shapes csrc/mlpw_*.hip is built for (tce_mlpw_supported)."""
    hl = list(mlp.hidden_layers)
    if not (mlp.dim_out == 1 and len(hl) == 2 and hl[0] == hl[1]
            and mlp.act_func_last_type is None
            and mlp.act_func_hidden_type in _ACT
            and mlp.dtype in (torch.float32, torch.float64)):
        return False
    if narrow_supported(mlp):
        return False
    esize = 4 if mlp.dtype == torch.float32 else 8
    return bool(_lib.load().tce_mlpw_supported(mlp.dim_in, hl[0], esize))


def supported(mlp):
    return narrow_supported(mlp) or wide_supported(mlp)


def _rows(x):
    """(tensor, env_stride, row_stride, T, R) describing x [..., din] in place
    when its rows are (env, step)-strided; a copy otherwise."""
    din = x.shape[-1]
    if x.dim() == 3 and x.stride(2) == 1:
        N, T, _ = x.shape
        return x, x.stride(0), x.stride(1), T, N * T
    x2 = x.reshape(-1, din)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    return x2, 0, x2.stride(0), x2.shape[0], x2.shape[0]


def _weights(mlp):
    ls = mlp.layers
    return [ptr(t) for t in (ls[0].weight, ls[0].bias, ls[1].weight,
                             ls[1].bias, ls[2].weight, ls[2].bias)]


def forward(mlp, x):
    """values [..., 1] without autograd (rollout / evaluation)."""
    if wide_supported(mlp):
        return _wide_forward(mlp, x)
    xs, es, rs, T, R = _rows(x)
    out = torch.empty(R, dtype=torch.float32, device=x.device)
    call("tce_mlp_critic_f32", ptr(xs), es, rs, T, R, mlp.dim_in,
         *_weights(mlp), _ACT[mlp.act_func_hidden_type], None, None, 0.0,
         ptr(out), None, None, None, 0, None, None, None, None, 0.0, 0.0, 0.0,
         0.0, 0.0, 0.0, 1.0, None, stream())
    return out.reshape(*x.shape[:-1], 1)


def _wide_ws(mlp, R, backward):
    """Workspace of the wide kernels, cached on the module (W2 images +, for
    the backward pass, H1 / dY2 / dY1 of all rows)."""
    H = mlp.hidden_layers[0]
    n = _lib.load().tce_mlpw_workspace_len(R, H, int(backward))
    ws = getattr(mlp, "_tce_wide_ws", None)
    dev = mlp.layers[0].weight.device
    if ws is None or ws.numel() < n or ws.dtype != mlp.dtype or ws.device != dev:
        ws = torch.empty(n, dtype=mlp.dtype, device=dev)
        mlp._tce_wide_ws = ws
    return ws


def _wide_forward(mlp, x):
    xs, es, rs, T, R = _rows(x)
    out = torch.empty(R, dtype=mlp.dtype, device=x.device)
    ws = _wide_ws(mlp, R, False)
    call("tce_mlpw_critic_" + sfx(mlp.dtype), ptr(xs), es, rs, T, R,
         mlp.dim_in, mlp.hidden_layers[0], *_weights(mlp),
         _ACT[mlp.act_func_hidden_type], None, None, 0.0, ptr(out), ptr(ws),
         None, None, None, 0, None, None, None, None, 0.0, 0.0, 0.0, 0.0, 0.0,
         0.0, 1.0, None, stream())
    return out.reshape(*x.shape[:-1], 1)


class WideEpochRunner:
    """EpochRunner for the wide / float64 value networks (csrc/mlpw_*.hip):
    same interface, two launches (+ the slab reduction) per epoch."""
    arith = "f32"

    def __init__(self, mlp, flat=None, arith="f32"):
        assert wide_supported(mlp) and arith == "f32"
        self.mlp = mlp
        lib = _lib.load()
        self.H = mlp.hidden_layers[0]
        self.P = lib.tce_mlpw_num_params(mlp.dim_in, self.H)
        dev = mlp.layers[0].weight.device
        if flat is None:
            flat = torch.zeros(self.P, dtype=mlp.dtype, device=dev)
        assert flat.numel() == self.P and flat.dtype == mlp.dtype
        self.flat = flat
        self.entry = "tce_mlpw_critic_" + sfx(mlp.dtype)
        self.partials = torch.empty(lib.tce_mlpw_grid(), self.P + 2,
                                    dtype=mlp.dtype, device=dev)
        off = 0
        self.params = list(mlp.parameters())
        self.views = []
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        assert off == self.P

    def epoch(self, states, returns, old_values, clip, max_workgroups=0,
              stats=None, adam=None, xchg=None, grad_scale=1.0):
        xs, es, rs, T, R = _rows(states)
        ret = returns.reshape(-1)
        ret = ret if ret.is_contiguous() else ret.contiguous()
        old = old_values.reshape(-1).contiguous() if clip > 0 else None
        if stats is None:
            stats = torch.zeros(4, dtype=self.mlp.dtype,
                                device=self.flat.device)
        if adam is not None:
            g = adam.param_groups[0]
            adam.host_step += 1
            adam._opt_called = True
            ad = (ptr(adam.flat_param), ptr(adam.m), ptr(adam.v),
                  ptr(adam.dev_state), float(g["lr"]), float(g["betas"][0]),
                  float(g["betas"][1]), float(g["eps"]),
                  float(g["weight_decay"]), float(adam.host_step))
        else:
            ad = (None, None, None, None, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)
        ws = _wide_ws(self.mlp, R, True)
        call(self.entry, ptr(xs), es, rs, T, R, self.mlp.dim_in, self.H,
             *_weights(self.mlp), _ACT[self.mlp.act_func_hidden_type],
             ptr(ret), ptr(old), float(clip), None, ptr(ws),
             ptr(self.partials), ptr(self.flat), ptr(stats),
             int(max_workgroups), *ad, float(grad_scale),
             None if xchg is None else xchg.handle, stream())
        for p, v in zip(self.params, self.views):
            p.grad = v
        return stats


    def epoch_minibatches(self, states, returns, old_values, clip, row_index,
                          num_minibatches, stats, adam, grad_clip=0.0,
                          max_workgroups=0, xchg=None, grad_scale=1.0):
        """One epoch in minibatches (tce_mlpw_critic_minibatch_*): see
        EpochRunner.epoch_minibatches."""
        xs, es, rs, T, R = _rows(states)
        assert row_index.dtype == torch.int64 and row_index.numel() == R \
            and row_index.is_contiguous()
        assert stats.shape == (num_minibatches, 4) and stats.is_contiguous()
        ret = returns.reshape(-1)
        ret = ret if ret.is_contiguous() else ret.contiguous()
        old = old_values.reshape(-1).contiguous() if clip > 0 else None
        g = adam.param_groups[0]
        first = adam.host_step + 1
        adam.host_step += num_minibatches
        adam._opt_called = True
        ws = _wide_ws(self.mlp, -(-R // num_minibatches), True)
        call(self.entry.replace("critic_", "critic_minibatch_"), ptr(xs), es,
             rs, T, R, self.mlp.dim_in, self.H, *_weights(self.mlp),
             _ACT[self.mlp.act_func_hidden_type], ptr(ret), ptr(old),
             float(clip), ptr(row_index), int(num_minibatches), ptr(ws),
             ptr(self.partials), ptr(self.flat), ptr(stats),
             int(max_workgroups), ptr(adam.flat_param), ptr(adam.m),
             ptr(adam.v), ptr(adam.dev_state), float(g["lr"]),
             float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
             float(g["weight_decay"]), float(first), float(grad_clip),
             float(grad_scale), None if xchg is None else xchg.handle,
             stream())
        for p, v in zip(self.params, self.views):
            p.grad = v
        return stats


def make_runner(mlp, flat=None, arith="f32"):
    """The epoch runner for a supported value network."""
    if wide_supported(mlp):
        return WideEpochRunner(mlp, flat, "f32")
    return EpochRunner(mlp, flat, arith)


class EpochRunner:
    """Holds the flat gradient buffer (p.grad are views of it) and the
    per-workgroup partial slabs for repeated critic epochs."""

    def __init__(self, mlp, flat=None, arith="f32"):
        """flat: gradient buffer to fill (the optimizer's flat gradient, in
        parameter order W1, b1, W2, b2, w3, b3); allocated if omitted.
        arith: "f32" (exact-fp32 matrix cores, csrc/mlp.hip), "bf16x3" (three-
        part bf16 operands on the bf16 matrix cores, csrc/mlpb.hip: 24-bit
        operands, fp32 range) or "f16x2" (split f16 operands on the f16 matrix
        cores, csrc/mlp16.hip: 22-bit operands inside the f16 range)."""
        assert narrow_supported(mlp)
        assert arith in ("f32", "f16x2", "bf16x3"), arith
        self.arith = arith
        self.entry = "tce_mlp_critic_" + arith
        self.mlp = mlp
        lib = _lib.load()
        self.P = lib.tce_mlp_critic_num_params(mlp.dim_in)
        dev = mlp.layers[0].weight.device
        if flat is None:
            flat = torch.zeros(self.P, dtype=torch.float32, device=dev)
        assert flat.numel() == self.P and flat.dtype == torch.float32
        self.flat = flat
        self.partials = torch.empty(lib.tce_mlp_critic_grid(), self.P + 2,
                                    dtype=torch.float32, device=dev)
        off = 0
        self.params = list(mlp.parameters())
        self.views = []
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        assert off == self.P

    def epoch(self, states, returns, old_values, clip, max_workgroups=0,
              stats=None, adam=None, xchg=None, grad_scale=1.0):
        """One full-batch forward + loss + backward; leaves the gradient in
        p.grad (views of the flat buffer) and returns stats = {mean loss,
        |grad|^2} as a device tensor [2] (``stats``: a ZEROED float32[2] to
        fill instead of a fresh one).  adam: a FlatAdam over the same
        parameters whose step (without clipping) is fused into the launch.
        xchg (a dist.Exchange; needs adam): the envs are sharded -- the
        reduction kernel adds the peers' gradients before Adam and applies
        grad_scale (1 / world) to the sum."""
        xs, es, rs, T, R = _rows(states)
        ret = returns.reshape(-1)
        ret = ret if ret.is_contiguous() else ret.contiguous()
        old = None
        if clip > 0:
            old = old_values.reshape(-1).contiguous()
        if stats is None:
            stats = torch.zeros(4, dtype=torch.float32, device=self.flat.device)
        if adam is not None:
            g = adam.param_groups[0]
            adam.host_step += 1
            adam._opt_called = True           # for LinearLR's order check
            ad = (ptr(adam.flat_param), ptr(adam.m), ptr(adam.v),
                  ptr(adam.dev_state), float(g["lr"]), float(g["betas"][0]),
                  float(g["betas"][1]), float(g["eps"]),
                  float(g["weight_decay"]), float(adam.host_step))
        else:
            ad = (None, None, None, None, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)
        call(self.entry, ptr(xs), es, rs, T, R, self.mlp.dim_in,
             *_weights(self.mlp), _ACT[self.mlp.act_func_hidden_type],
             ptr(ret), ptr(old), float(clip), None, ptr(self.partials),
             ptr(self.flat), ptr(stats), int(max_workgroups), *ad,
             float(grad_scale), None if xchg is None else xchg.handle,
             stream())
        for p, v in zip(self.params, self.views):
            p.grad = v
        return stats


    def epoch_minibatches(self, states, returns, old_values, clip, row_index,
                          num_minibatches, stats, adam, grad_clip=0.0,
                          max_workgroups=0, xchg=None, grad_scale=1.0):
        """One critic epoch of ``num_minibatches`` optimizer steps
        (temporal_correlated_agent.py:343-366) as ONE C call
        (tce_mlp_critic_minibatch_f32): ``row_index`` int64 [R] on the device =
        the epoch's permutation (util_data_structure.py:378-391), cut like
        np.array_split; the kernels read the gathered rows in place.  stats
        [num_minibatches, 4] zeroed; adam: the FlatAdam whose steps are taken
        (grad_clip > 0: clip + Adam as one launch per piece)."""
        if self.arith != "f32":
            raise NotImplementedError(
                "minibatched critic epochs run on the exact-fp32 kernel only "
                "(critic_arith=%s)" % self.arith)
        xs, es, rs, T, R = _rows(states)
        assert row_index.dtype == torch.int64 and row_index.numel() == R \
            and row_index.is_contiguous()
        assert stats.shape == (num_minibatches, 4) and stats.is_contiguous()
        ret = returns.reshape(-1)
        ret = ret if ret.is_contiguous() else ret.contiguous()
        old = old_values.reshape(-1).contiguous() if clip > 0 else None
        g = adam.param_groups[0]
        first = adam.host_step + 1
        adam.host_step += num_minibatches
        adam._opt_called = True
        call("tce_mlp_critic_minibatch_f32", ptr(xs), es, rs, T, R,
             self.mlp.dim_in, *_weights(self.mlp),
             _ACT[self.mlp.act_func_hidden_type], ptr(ret), ptr(old),
             float(clip), ptr(row_index), int(num_minibatches),
             ptr(self.partials), ptr(self.flat), ptr(stats),
             int(max_workgroups), ptr(adam.flat_param), ptr(adam.m),
             ptr(adam.v), ptr(adam.dev_state), float(g["lr"]),
             float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
             float(g["weight_decay"]), float(first), float(grad_clip),
             float(grad_scale), None if xchg is None else xchg.handle,
             stream())
        for p, v in zip(self.params, self.views):
            p.grad = v
        return stats


# ---------------------------------------------------------------------------
# hidden layers of a wider-output net (policy mean net) on the same kernels
# ---------------------------------------------------------------------------
def hidden_supported(mlp, x):
    return (mlp.dtype == torch.float32 and list(mlp.hidden_layers) == [128, 128]
            and 1 <= mlp.dim_in <= 40 and mlp.act_func_hidden_type in _ACT
            and x.is_cuda and x.dtype == torch.float32)


class _Hidden2(torch.autograd.Function):
    """h2 = act(W2 act(W1 x + b1) + b2) for x [R, D_in]; backward recomputes the
    forward inside the fused kernel and returns dW1, db1, dW2, db2."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act):
        xs, es, rs, T, R = _rows(x)
        out = torch.empty(R, 128, dtype=torch.float32, device=x.device)
        call("tce_mlp_hidden_f32", ptr(xs), es, rs, T, R, x.shape[-1], ptr(w1),
             ptr(b1), ptr(w2), ptr(b2), act, None, ptr(out), None, None, None,
             stream())
        ctx.save_for_backward(xs, w1, b1, w2, b2)
        ctx.geom = (es, rs, T, R, x.shape[-1], act)
        return out.reshape(*x.shape[:-1], 128)

    @staticmethod
    def backward(ctx, g):
        xs, w1, b1, w2, b2 = ctx.saved_tensors
        es, rs, T, R, din, act = ctx.geom
        lib = _lib.load()
        P = lib.tce_mlp_critic_num_params(din)
        dev = g.device
        g = g.reshape(R, 128)
        g = g if g.is_contiguous() else g.contiguous()
        partials = torch.empty(min(lib.tce_mlp_critic_grid(), (R + 63) // 64),
                               P + 2, dtype=torch.float32, device=dev)
        grad = torch.empty(P, dtype=torch.float32, device=dev)
        stats = torch.zeros(2, dtype=torch.float32, device=dev)
        call("tce_mlp_hidden_f32", ptr(xs), es, rs, T, R, din, ptr(w1), ptr(b1),
             ptr(w2), ptr(b2), act, ptr(g), None, ptr(partials), ptr(grad),
             ptr(stats), stream())
        o1, o2, o3 = 128 * din, 128 * din + 128, 128 * din + 128 + 128 * 128
        return (None, grad[:o1].view(128, din), grad[o1:o2],
                grad[o2:o3].view(128, 128), grad[o3:o3 + 128], None)


def hidden_forward(mlp, x):
    """Activations of the second hidden layer [..., 128] (differentiable w.r.t.
    the four hidden-layer parameters, not w.r.t. x)."""
    ls = mlp.layers
    return _Hidden2.apply(x, ls[0].weight, ls[0].bias, ls[1].weight,
                          ls[1].bias, _ACT[mlp.act_func_hidden_type])
