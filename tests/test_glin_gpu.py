"""The generic dense layer of csrc/glin.hip (``mlp_ops.HipLinear``) against
torch's ``F.linear`` under autograd: forward, input gradient, weight gradient
(rows split over the chip), bias gradient -- float32 and float64, ragged tiles,
strided rows, the K (K + 1) / 2 outputs of a contextual covariance head
(mprl/rl/policy/abstract_policy.py:96-109) and the row counts of a critic
batch.  Reference: ``MLP.forward`` (mprl/util/util_nn.py:225-246)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(1, 1, 1), (17, 10, 48), (64, 16, 64), (65, 17, 65),
          (4096, 40, 300), (300, 64, 300), (513, 300, 2080), (70001, 48, 64),
          (33, 4096, 3), (3, 5, 4096)]


@pytest.mark.parametrize("R,din,dout", SHAPES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_hip_linear_equals_torch(R, din, dout, dtype):
    from tce_rl_amd.mlp_ops import HipLinear
    g = torch.Generator(device="cuda").manual_seed(R + din)
    full = torch.randn(R, din + 7, device="cuda", dtype=dtype, generator=g)
    x = full[:, :din]                                   # strided rows
    w = torch.randn(dout, din, device="cuda", dtype=dtype, generator=g) / din ** 0.5
    b = torch.randn(dout, device="cuda", dtype=dtype, generator=g)
    up = torch.randn(R, dout, device="cuda", dtype=dtype, generator=g)
    assert HipLinear.supported(x, w)
    out = {}
    for name, fn, dt in (("hip", HipLinear.apply, dtype),
                         ("ref", F.linear, torch.float64)):
        xx = x.to(dt).clone().requires_grad_(True)
        ww, bb = w.to(dt).clone().requires_grad_(True), \
            b.to(dt).clone().requires_grad_(True)
        y = fn(xx, ww, bb)
        (y * up.to(dt)).sum().backward()
        out[name] = (y.detach(), xx.grad, ww.grad, bb.grad)
    # float32: the rounding of a din- / dout- / R-term sum; float64: exact to
    # the order of summation
    for a, r, terms in zip(out["hip"], out["ref"], (din, dout, R, R)):
        tol = (3e-7 * terms ** 0.5 + 1e-6) if dtype == torch.float32 else 1e-12
        scale = max(float(r.abs().max()), 1e-30)
        err = float((a.double() - r).abs().max())
        assert err <= tol * scale * 8, (a.shape, err, scale)


def test_hip_linear_three_dim_input_no_bias_and_gradcheck():
    from tce_rl_amd.mlp_ops import HipLinear
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(5, 7, 9, device="cuda", dtype=torch.float64, generator=g,
                    ).requires_grad_(True)
    w = torch.randn(11, 9, device="cuda", dtype=torch.float64, generator=g,
                    ).requires_grad_(True)
    y = HipLinear.apply(x, w, None)
    assert y.shape == (5, 7, 11)
    torch.testing.assert_close(y, F.linear(x, w), rtol=1e-12, atol=1e-12)
    assert torch.autograd.gradcheck(
        lambda a, b: HipLinear.apply(a, b, None), (x, w), eps=1e-6, atol=1e-6)
    # weight gradient is repeatable bit for bit (fixed split order)
    up = torch.randn(3000, 11, device="cuda", dtype=torch.float64, generator=g)
    xs = torch.randn(3000, 9, device="cuda", dtype=torch.float64, generator=g)
    gs = []
    for _ in range(2):
        ww = w.detach().clone().requires_grad_(True)
        (HipLinear.apply(xs, ww, None) * up).sum().backward()
        gs.append(ww.grad.clone())
    assert torch.equal(gs[0], gs[1])


@pytest.mark.parametrize("hidden,dout", [([64, 64], 300), ([48, 48], 3),
                                         ([512, 512, 512], 24), ([200], 7)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_uncovered_net_shapes_stay_off_the_library(hidden, dout, dtype):
    """MLP widths outside the fused families (the
    contextual covariance head 40 -> 64 -> 64 -> 300, odd widths, three hidden
    layers) run layer by layer on the generic kernel under autograd -- no
    library GEMM, no warning -- and equal the torch float64 reference."""
    import warnings
    from tce_rl_amd import mlp_ops
    from tce_rl_amd.nn import MLP
    torch.manual_seed(len(hidden) + dout)
    mlp = MLP("m", 40, dout, hidden, "orthogonal", 0.7, "tanh", None, dtype,
              torch.device("cuda"))
    x = torch.randn(777, 40, device="cuda", dtype=dtype)
    up = torch.randn(777, dout, device="cuda", dtype=dtype)
    mlp_ops.LIBRARY_CALLS.clear()
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        y = mlp(x)
        (y * up).sum().backward()
    assert not mlp_ops.LIBRARY_CALLS
    ws = [p.detach().double().requires_grad_(True) for p in mlp.parameters()]
    h = x.double()
    for i in range(len(hidden)):
        h = torch.tanh(F.linear(h, ws[2 * i], ws[2 * i + 1]))
    yr = F.linear(h, ws[-2], ws[-1])
    (yr * up.double()).sum().backward()
    tol = 2e-5 if dtype == torch.float32 else 1e-11
    torch.testing.assert_close(y.double(), yr, rtol=tol, atol=tol)
    for p, w in zip(mlp.parameters(), ws):
        s = max(float(w.grad.abs().max()), 1.0)
        torch.testing.assert_close(p.grad.double(), w.grad, rtol=tol,
                                   atol=tol * s)
