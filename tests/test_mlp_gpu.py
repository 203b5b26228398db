"""GPU parity of the fused fp32-MFMA critic kernel against a plain PyTorch
fp32/fp64 reference of the same op (forward, value loss, all gradients)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ACTS = {"tanh": torch.tanh, "relu": F.relu, "leaky_relu": F.leaky_relu,
        "softplus": F.softplus}


def make(din, act, seed):
    from tce_rl_amd.nn import MLP
    torch.manual_seed(seed)
    return MLP("ValueFunction", din, 1, [128, 128], "orthogonal", 1.0, act,
               None, torch.float32, torch.device("cuda"))


def torch_ref(mlp, x, ret, old, clip, dtype):
    ws = [p.detach().to(dtype).requires_grad_(True) for p in mlp.parameters()]
    h = x.to(dtype)
    act = ACTS[mlp.act_func_hidden_type]
    h = act(F.linear(h, ws[0], ws[1]))
    h = act(F.linear(h, ws[2], ws[3]))
    v = F.linear(h, ws[4], ws[5]).squeeze(-1)
    r, o = ret.to(dtype), old.to(dtype)
    loss = (r - v).pow(2)
    if clip > 0:
        vc = o + (v - o).clamp(-clip, clip)
        loss = torch.max(loss, (vc - r).pow(2))
    loss = loss.mean()
    loss.backward()
    return v.detach(), loss.detach(), [w.grad for w in ws]


@pytest.mark.parametrize("act", ["relu", "tanh", "leaky_relu", "softplus"])
@pytest.mark.parametrize("din,N,T", [(40, 7, 33), (21, 5, 64), (32, 3, 1),
                                     (17, 130, 10), (24, 9, 21), (25, 4, 70),
                                     (16, 3, 40), (1, 6, 11)])
def test_fused_critic_epoch_vs_torch(act, din, N, T):
    from tce_rl_amd import critic_ops
    mlp = make(din, act, 0)
    D = din + 8
    g = torch.Generator(device="cuda").manual_seed(1)
    full = torch.randn(N, T + 1, D, device="cuda", generator=g)
    states = full[:, :-1]                         # strided view, like the agent
    ret = torch.randn(N, T, device="cuda", generator=g) * 3
    old = torch.randn(N, T, device="cuda", generator=g)
    x = states[..., :din]
    for clip in (0.0, 0.7):
        v64, l64, g64 = torch_ref(mlp, x.reshape(-1, din), ret.reshape(-1),
                                  old.reshape(-1), clip, torch.float64)
        v32, l32, g32 = torch_ref(mlp, x.reshape(-1, din), ret.reshape(-1),
                                  old.reshape(-1), clip, torch.float32)
        vals = critic_ops.forward(mlp, x)
        assert vals.shape == (N, T, 1)
        err = (vals.reshape(-1).double() - v64).abs().max()
        ref_err = (v32.double() - v64).abs().max()
        assert err <= 4 * ref_err + 1e-6, (err, ref_err)
        run = critic_ops.EpochRunner(mlp)
        stats = run.epoch(x, ret, old, clip).cpu()
        assert abs(stats[0].item() - l64.item()) <= 1e-5 * abs(l64.item()) + 1e-6
        gn2 = sum((gg.double() ** 2).sum() for gg in g64).item()
        assert abs(stats[1].item() - gn2) <= 1e-4 * gn2 + 1e-9
        for p, a, b in zip(mlp.parameters(), g64, g32):
            e = (p.grad.double() - a).abs().max().item()
            e32 = (b.double() - a).abs().max().item()
            scale = a.abs().max().item()
            # fp32 rounding level of the gradient magnitude (summation order
            # differs from the library GEMMs)
            assert e <= 4 * e32 + 1e-5 * scale + 1e-7, (p.shape, e, e32, scale)


def test_fused_critic_c2_shape_matches_library_path():
    """BASELINE C2 rows (4096 x 500 x 48, D_in 40): fused values == library
    GEMM values to fp32 rounding; gradients agree with torch autograd."""
    from tce_rl_amd import critic_ops, mlp_ops
    mlp = make(40, "relu", 3)
    g = torch.Generator(device="cuda").manual_seed(2)
    full = torch.randn(4096, 501, 48, device="cuda", generator=g)
    x = full[:, :-1, :40]
    v_lib = mlp_ops.forward(mlp, x)
    v = critic_ops.forward(mlp, x)
    torch.testing.assert_close(v, v_lib, rtol=1e-4, atol=1e-5)
    ret = torch.randn(4096, 500, device="cuda", generator=g)
    run = critic_ops.EpochRunner(mlp)
    stats = run.epoch(x, ret, ret, 0.0)
    mine = [p.grad.clone() for p in mlp.parameters()]
    for p in mlp.parameters():
        p.grad = None
    loss = (ret - mlp_ops.forward(mlp, x).squeeze(-1)).pow(2).mean()
    loss.backward()
    torch.testing.assert_close(stats[0], loss.detach(), rtol=1e-5, atol=1e-6)
    for a, p in zip(mine, mlp.parameters()):
        torch.testing.assert_close(a, p.grad, rtol=2e-3, atol=2e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("rows", [100, (1 << 15) + 77, 70000])
def test_library_path_split_k_weight_gradient(dtype, rows):
    """mlp_ops._Linear (split-K weight gradient) == plain F.linear autograd."""
    from tce_rl_amd import mlp_ops
    g = torch.Generator(device="cuda").manual_seed(rows)
    x = torch.randn(rows, 48, device="cuda", dtype=dtype, generator=g)[:, :27]
    w = torch.randn(64, 27, device="cuda", dtype=dtype, generator=g)
    b = torch.randn(64, device="cuda", dtype=dtype, generator=g)
    outs = []
    for fn in (mlp_ops._Linear.apply, F.linear):
        xx = x.clone().requires_grad_(True)
        ww, bb = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = fn(xx, ww, bb)
        (y.tanh().pow(2).mean()).backward()
        outs.append((y.detach(), xx.grad, ww.grad, bb.grad))
    tol = 2e-4 if dtype == torch.float32 else 1e-11
    for a, r in zip(*outs):
        torch.testing.assert_close(a, r, rtol=tol, atol=tol * r.abs().max().item())


@pytest.mark.parametrize("act", ["relu", "tanh", "leaky_relu", "softplus"])
@pytest.mark.parametrize("din,rows,dout", [(39, 300, 24), (27, 4096, 63),
                                           (7, 1, 3), (40, 65, 1)])
def test_hidden_layers_on_the_fused_kernels(act, din, rows, dout):
    """Policy-sized nets: both hidden layers (forward + backward) run in the
    fused MFMA kernels (tce_mlp_hidden_f32), the output layer is a GEMM."""
    from tce_rl_amd import critic_ops, mlp_ops
    from tce_rl_amd.nn import MLP
    torch.manual_seed(din)
    mlp = MLP("m", din, dout, [128, 128], "orthogonal", 0.01, act, None,
              torch.float32, torch.device("cuda"))
    g = torch.Generator(device="cuda").manual_seed(rows)
    x = torch.randn(rows, din + 9, device="cuda", generator=g)[:, :din]
    up = torch.randn(rows, dout, device="cuda", generator=g)
    assert critic_ops.hidden_supported(mlp, x)
    y = mlp_ops.forward(mlp, x)
    (y * up).sum().backward()
    got = [y.detach()] + [p.grad.clone() for p in mlp.parameters()]
    ref = {}
    for dtype in (torch.float64, torch.float32):
        ws = [p.detach().to(dtype).requires_grad_(True)
              for p in mlp.parameters()]
        h = x.to(dtype)
        h = ACTS[act](F.linear(h, ws[0], ws[1]))
        h = ACTS[act](F.linear(h, ws[2], ws[3]))
        yy = F.linear(h, ws[4], ws[5])
        (yy * up.to(dtype)).sum().backward()
        ref[dtype] = [yy.detach()] + [w.grad for w in ws]
    for a, r64, r32 in zip(got, ref[torch.float64], ref[torch.float32]):
        e = (a.double() - r64).abs().max().item()
        e32 = (r32.double() - r64).abs().max().item()
        scale = r64.abs().max().item()
        assert e <= 4 * e32 + 1e-5 * scale + 1e-7, (a.shape, e, e32, scale)


def test_library_gemm_fallback_is_counted_and_announced_once():
    """a net no hand-written kernel covers used to take
    F.linear + autograd silently.  Since round 6 the generic dense layer of
    csrc/glin.hip takes every float32 / float64 shape up to 4096 wide; what is
    left for the library (a wider layer) still computes, is counted per shape
    and a RuntimeWarning names the shape once."""
    import warnings
    from tce_rl_amd import mlp_ops
    from tce_rl_amd.nn import MLP
    mlp_ops.LIBRARY_CALLS.clear()
    mlp_ops._warned.clear()
    odd = MLP("odd", 10, 3, [48, 48], "orthogonal", 1.0, "tanh", None,
              torch.float32, torch.device("cuda"))
    x = torch.randn(17, 10, device="cuda")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        y = odd(x)
    assert y.shape == (17, 3) and y.requires_grad
    assert not w and not mlp_ops.LIBRARY_CALLS           # hand-written now
    huge = MLP("huge", 10, 3, [4100], "orthogonal", 1.0, "tanh", None,
               torch.float32, torch.device("cuda"))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        y = huge(x)
        huge(x)
    assert y.shape == (17, 3) and y.requires_grad
    assert len([m for m in w if "library GEMMs" in str(m.message)]) == 1
    key = ("library", "torch.float32", 10, 4100, 3)
    assert mlp_ops.LIBRARY_CALLS == {key: 4}             # 2 layers x 2 calls
    # a covered shape under no_grad leaves the counter alone
    good = MLP("good", 39, 1, [32, 32], "orthogonal", 1.0, "relu", None,
               torch.float32, torch.device("cuda"))
    with torch.no_grad():
        good(torch.randn(64, 39, device="cuda"))
    assert mlp_ops.LIBRARY_CALLS == {key: 4}
    mlp_ops.LIBRARY_CALLS.clear()
