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
