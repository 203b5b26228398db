"""MLP forward / backward dispatch.

Library-GEMM path: the dense layers go through torch.nn.functional.linear
(hipBLASLt / rocBLAS on ROCm -- plain library GEMMs) with torch autograd.  It is
the reference-precision path for every MLP shape; the fused MFMA kernels of
csrc/mlp.hip replace it for the critic's large-batch epochs (critic_ops).
"""
import torch
import torch.nn.functional as F

_ACT = {"tanh": torch.tanh, "relu": F.relu, "leaky_relu": F.leaky_relu,
        "softplus": F.softplus}


def forward(mlp, x):
    if not x.is_cuda:
        raise RuntimeError("tce_rl_amd MLPs run on a HIP device only")
    from . import critic_ops
    if not torch.is_grad_enabled() and critic_ops.supported(mlp) \
            and x.numel() >= 4096 * mlp.dim_in:
        return critic_ops.forward(mlp, x)         # fused MFMA forward
    layers = mlp.layers
    act = _ACT[mlp.act_func_hidden_type]
    for i in range(len(mlp.hidden_layers)):
        x = act(F.linear(x, layers[i].weight, layers[i].bias))
    x = F.linear(x, layers[-1].weight, layers[-1].bias)
    if mlp.act_func_last_type is not None:
        x = _ACT[mlp.act_func_last_type](x)
    return x
