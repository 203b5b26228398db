"""GPU parity of the split-f16 critic kernel (csrc/mlp16.hip) against a plain
PyTorch fp64 reference of the same op, held to the SAME bound as the exact-fp32
kernel (tests/test_mlp_gpu.py): the error may not exceed a few times the error
of the fp32 PyTorch reference itself."""
import pytest
import torch

from test_mlp_gpu import make, torch_ref

pytestmark = pytest.mark.gpu


def run_case(act, din, N, T, seed=0, scale_x=1.0, scale_ret=3.0):
    from tce_rl_amd import critic_ops
    mlp = make(din, act, seed)
    D = din + 8
    g = torch.Generator(device="cuda").manual_seed(seed + 1)
    full = torch.randn(N, T + 1, D, device="cuda", generator=g) * scale_x
    states = full[:, :-1]
    ret = torch.randn(N, T, device="cuda", generator=g) * scale_ret
    old = torch.randn(N, T, device="cuda", generator=g)
    x = states[..., :din]
    out = []
    for clip in (0.0, 0.7):
        v64, l64, g64 = torch_ref(mlp, x.reshape(-1, din), ret.reshape(-1),
                                  old.reshape(-1), clip, torch.float64)
        v32, l32, g32 = torch_ref(mlp, x.reshape(-1, din), ret.reshape(-1),
                                  old.reshape(-1), clip, torch.float32)
        run = critic_ops.EpochRunner(mlp, arith="f16x2")
        stats = run.epoch(x, ret, old, clip).cpu()
        out.append((stats, l64, g64, g32, [p.grad.clone() for p in mlp.parameters()]))
    return out


@pytest.mark.parametrize("act", ["relu", "tanh", "leaky_relu", "softplus"])
@pytest.mark.parametrize("din,N,T", [(40, 7, 33), (21, 5, 64), (32, 3, 1),
                                     (17, 130, 10), (31, 9, 21), (39, 4, 70),
                                     (33, 3, 40), (1, 6, 11)])
def test_split_f16_epoch_vs_torch(act, din, N, T):
    for stats, l64, g64, g32, grads in run_case(act, din, N, T):
        assert abs(stats[0].item() - l64.item()) <= 1e-5 * abs(l64.item()) + 1e-6
        gn2 = sum((gg.double() ** 2).sum() for gg in g64).item()
        assert abs(stats[1].item() - gn2) <= 1e-4 * gn2 + 1e-9
        for gk, a, b in zip(grads, g64, g32):
            e = (gk.double() - a).abs().max().item()
            e32 = (b.double() - a).abs().max().item()
            scale = a.abs().max().item()
            assert e <= 4 * e32 + 1e-5 * scale + 1e-7, (gk.shape, e, e32, scale)


@pytest.mark.parametrize("scale_x,scale_ret", [(1e-3, 1e-3), (30.0, 500.0), (1.0, 1e-4)])
def test_split_f16_operand_ranges(scale_x, scale_ret):
    """Tiny and large operands: the split keeps 22 bits down to the f16
    subnormal range and the backward scale keeps dL/dv inside it."""
    for stats, l64, g64, g32, grads in run_case("relu", 40, 33, 50, seed=3,
                                                scale_x=scale_x, scale_ret=scale_ret):
        assert abs(stats[0].item() - l64.item()) <= 1e-5 * abs(l64.item()) + 1e-12
        for gk, a, b in zip(grads, g64, g32):
            e = (gk.double() - a).abs().max().item()
            e32 = (b.double() - a).abs().max().item()
            scale = a.abs().max().item()
            assert e <= 4 * e32 + 2e-5 * scale, (gk.shape, e, e32, scale)


def test_split_f16_c2_shape_matches_fp32_kernel():
    """BASELINE C2 rows (4096 x 500, D_in 40): loss and gradient of the split-f16
    kernel == the exact-fp32 kernel to fp32 summation-order level."""
    from tce_rl_amd import critic_ops
    mlp = make(40, "relu", 5)
    g = torch.Generator(device="cuda").manual_seed(2)
    full = torch.randn(4096, 501, 48, device="cuda", generator=g)
    x = full[:, :-1, :40]
    ret = torch.randn(4096, 500, device="cuda", generator=g)
    a = critic_ops.EpochRunner(mlp, arith="f32")
    sa = a.epoch(x, ret, ret, 0.0).cpu()
    ga = a.flat.clone()
    b = critic_ops.EpochRunner(mlp, arith="f16x2")
    sb = b.epoch(x, ret, ret, 0.0).cpu()
    gb = b.flat.clone()
    assert abs(sa[0] - sb[0]).item() <= 2e-6 * abs(sa[0]).item()
    assert (ga - gb).abs().max().item() <= 2e-5 * ga.abs().max().item()
    assert ((ga - gb).norm() / ga.norm()).item() <= 5e-6


def test_split_f16_values_output_and_overflow_is_reported():
    """The launch can also emit the values; operands beyond the f16 range do not
    pass silently: the loss comes back non-finite (the agent's NaN check)."""
    from tce_rl_amd import _lib, critic_ops
    from tce_rl_amd._lib import call, ptr, stream
    mlp = make(24, "tanh", 2)
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.randn(1000, 24, device="cuda", generator=g)
    ret = torch.randn(1000, device="cuda", generator=g)
    lib = _lib.load()
    P = lib.tce_mlp_critic_num_params(24)
    partials = torch.empty(lib.tce_mlp_critic_grid(), P + 2, device="cuda")
    grad, stats = torch.empty(P, device="cuda"), torch.zeros(2, device="cuda")
    vals = torch.empty(1000, device="cuda")
    ws = [ptr(p) for p in mlp.parameters()]
    call("tce_mlp_critic_f16x2", ptr(x), 0, 24, 1000, 1000, 24, *ws, 0, ptr(ret),
         None, 0.0, ptr(vals), ptr(partials), ptr(grad), ptr(stats), 0, None,
         None, None, None, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, None, stream())
    ref = critic_ops.forward(mlp, x.expand(5, 1000, 24).contiguous())[0, :, 0]
    torch.testing.assert_close(vals, ref, rtol=1e-5, atol=1e-6)
    stats.zero_()
    big = x * 1e6                                   # |x| > 65504
    call("tce_mlp_critic_f16x2", ptr(big), 0, 24, 1000, 1000, 24, *ws, 0,
         ptr(ret), None, 0.0, None, ptr(partials), ptr(grad), ptr(stats), 0,
         None, None, None, None, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, None,
         stream())
    assert not torch.isfinite(stats[0]).item()
