"""GPU parity of the wide / float64 fused critic kernels (csrc/mlpw_*.hip:
D_in -> 256 -> 256 -> 1 in fp32 and fp64, D_in -> 128 -> 128 -> 1 in fp64; the
critics of the box-pushing and table-tennis configs) against a plain PyTorch
reference of the same op: forward values, value loss, all six gradients, the
fused Adam step."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ACTS = {"tanh": torch.tanh, "relu": F.relu, "leaky_relu": F.leaky_relu,
        "softplus": F.softplus}


def make(din, hidden, act, dtype, seed):
    from tce_rl_amd.nn import MLP
    torch.manual_seed(seed)
    return MLP("ValueFunction", din, 1, [hidden, hidden], "orthogonal", 1.0, act,
               None, dtype, torch.device("cuda"))


def torch_ref(mlp, x, ret, old, clip, dtype):
    ws = [p.detach().to(dtype).requires_grad_(True) for p in mlp.parameters()]
    act = ACTS[mlp.act_func_hidden_type]
    h = act(F.linear(x.to(dtype), ws[0], ws[1]))
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


SHAPES = [(21, 5, 64), (40, 7, 33), (24, 3, 1), (1, 6, 11), (17, 130, 10),
          (33, 2, 300)]


@pytest.mark.parametrize("act", ["leaky_relu", "tanh", "relu", "softplus"])
@pytest.mark.parametrize("hidden,dtype", [(256, torch.float32),
                                          (256, torch.float64),
                                          (128, torch.float64)])
@pytest.mark.parametrize("din,N,T", SHAPES)
def test_wide_critic_epoch_vs_torch(act, hidden, dtype, din, N, T):
    from tce_rl_amd import critic_ops
    if dtype == torch.float64 and hidden == 256 and din > 24:
        mlp = make(din, hidden, act, dtype, 0)
        assert not critic_ops.supported(mlp)      # W1 image does not fit: library path
        return
    mlp = make(din, hidden, act, dtype, 0)
    assert critic_ops.wide_supported(mlp) and critic_ops.supported(mlp)
    D = din + 8
    g = torch.Generator(device="cuda").manual_seed(1)
    full = torch.randn(N, T + 1, D, device="cuda", generator=g, dtype=dtype)
    x = full[:, :-1, :din]                        # strided view, like the agent
    ret = torch.randn(N, T, device="cuda", generator=g, dtype=dtype) * 3
    old = torch.randn(N, T, device="cuda", generator=g, dtype=dtype)
    for clip in (0.0, 0.7):
        v64, l64, g64 = torch_ref(mlp, x.reshape(-1, din), ret.reshape(-1),
                                  old.reshape(-1), clip, torch.float64)
        vals = critic_ops.forward(mlp, x)
        assert vals.shape == (N, T, 1) and vals.dtype == dtype
        run = critic_ops.make_runner(mlp)
        stats = run.epoch(x, ret, old, clip).cpu()
        if dtype == torch.float64:
            torch.testing.assert_close(vals.reshape(-1), v64, rtol=1e-11,
                                       atol=1e-12)
            assert abs(stats[0].item() - l64.item()) <= 1e-12 * abs(l64.item())
            gn2 = sum((gg ** 2).sum() for gg in g64).item()
            assert abs(stats[1].item() - gn2) <= 1e-11 * gn2
            for p, a in zip(mlp.parameters(), g64):
                torch.testing.assert_close(
                    p.grad, a, rtol=1e-10, atol=1e-12 * a.abs().max().item())
        else:
            v32, l32, g32 = torch_ref(mlp, x.reshape(-1, din), ret.reshape(-1),
                                      old.reshape(-1), clip, torch.float32)
            err = (vals.reshape(-1).double() - v64).abs().max()
            ref_err = (v32.double() - v64).abs().max()
            assert err <= 4 * ref_err + 1e-6, (err, ref_err)
            assert abs(stats[0].item() - l64.item()) <= \
                1e-5 * abs(l64.item()) + 1e-6
            gn2 = sum((gg.double() ** 2).sum() for gg in g64).item()
            assert abs(stats[1].item() - gn2) <= 1e-4 * gn2 + 1e-9
            for p, a, b in zip(mlp.parameters(), g64, g32):
                e = (p.grad.double() - a).abs().max().item()
                e32 = (b.double() - a).abs().max().item()
                scale = a.abs().max().item()
                assert e <= 4 * e32 + 1e-5 * scale + 1e-7, \
                    (p.shape, e, e32, scale)


@pytest.mark.parametrize("hidden,dtype,din", [(256, torch.float32, 21),
                                              (256, torch.float64, 21),
                                              (128, torch.float64, 40)])
def test_wide_critic_many_rows_and_fused_adam(hidden, dtype, din):
    """More rows than one pass of the persistent grid (several tiles / row
    chunks per workgroup), a workgroup limit, and the Adam step fused into the
    slab reduction == torch.optim.Adam on the torch gradients."""
    from tce_rl_amd import critic_ops
    from tce_rl_amd.optim import FlatAdam
    mlp = make(din, hidden, "leaky_relu", dtype, 3)
    N, T = 700, 100                                # 70 000 rows
    g = torch.Generator(device="cuda").manual_seed(2)
    full = torch.randn(N, T + 1, din + 14, device="cuda", generator=g,
                       dtype=dtype)
    x = full[:, :-1, :din]
    ret = torch.randn(N, T, device="cuda", generator=g, dtype=dtype)
    _, l64, g64 = torch_ref(mlp, x.reshape(-1, din), ret.reshape(-1),
                            ret.reshape(-1), 0.0, torch.float64)
    ref_params = [p.detach().double().clone().requires_grad_(True)
                  for p in mlp.parameters()]
    ref_opt = torch.optim.Adam(ref_params, lr=3e-3, weight_decay=1e-4)
    for p, gr in zip(ref_params, g64):
        p.grad = gr.clone()
    ref_opt.step()
    opt = FlatAdam(list(mlp.parameters()), lr=3e-3, weight_decay=1e-4)
    run = critic_ops.make_runner(mlp, opt.flat_grad)
    opt.bind_grads()
    for wg in (0, 37):
        stats = run.epoch(x, ret, ret, 0.0, max_workgroups=wg)
        tol = 1e-10 if dtype == torch.float64 else 2e-5
        assert abs(stats[0].item() - l64.item()) <= tol * abs(l64.item())
        for p, a in zip(mlp.parameters(), g64):
            torch.testing.assert_close(p.grad.double(), a, rtol=50 * tol,
                                       atol=tol * a.abs().max().item())
    stats = torch.zeros(2, dtype=dtype, device="cuda")
    run.epoch(x, ret, ret, 0.0, stats=stats, adam=opt)
    ptol = 1e-9 if dtype == torch.float64 else 2e-5
    for p, r in zip(mlp.parameters(), ref_params):
        torch.testing.assert_close(p.detach().double(), r.detach(), rtol=ptol,
                                   atol=ptol)
    assert opt.host_step == 1


def test_wide_critic_argument_checks():
    from tce_rl_amd import _lib
    lib = _lib.load()
    assert lib.tce_mlpw_supported(21, 256, 4) and lib.tce_mlpw_supported(21, 256, 8)
    assert lib.tce_mlpw_supported(40, 128, 8)
    assert not lib.tce_mlpw_supported(40, 256, 8)
    assert not lib.tce_mlpw_supported(21, 192, 4)
    assert not lib.tce_mlpw_supported(41, 256, 4)
    x = torch.zeros(64, 21, device="cuda")
    w = torch.zeros(256 * 256, device="cuda")
    out = torch.zeros(64, device="cuda")
    ws = torch.zeros(lib.tce_mlpw_workspace_len(64, 256, 0), device="cuda")
    p = lambda t: t.data_ptr()
    base = [p(x), 0, 21, 64, 64, 21, 256, p(w), p(w), p(w), p(w), p(w), p(w), 1,
            None, None, 0.0, p(out), p(ws), None, None, None, 0, None, None,
            None, None, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, None, 0]
    _lib.call("tce_mlpw_critic_f32", *base)
    for idx, bad in ((6, 192), (13, 9), (17, None), (5, 0)):
        args = list(base)
        args[idx] = bad
        with pytest.raises(RuntimeError, match="mlpw_critic"):
            _lib.call("tce_mlpw_critic_f32", *args)
