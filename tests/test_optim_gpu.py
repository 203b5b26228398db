"""Flat Adam kernel (csrc/optim.hip, tce_rl_amd/optim.py) against
torch.optim.Adam + the reference's grad_norm_clip rule, and checkpoint
compatibility of its state_dict with torch's."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(dtype, seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(128, 39), (128,), (128, 128), (128,), (24, 128), (24,), (300,)]
    return [torch.randn(*s, generator=g, dtype=dtype) for s in shapes]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("wd,clip", [(0.0, 0.0), (1e-3, 0.0), (0.0, 0.5),
                                     (1e-2, 5.0)])
@pytest.mark.parametrize("once", [False, True])
def test_flat_adam_matches_torch_adam(dtype, wd, clip, once):
    """once: the single-launch form (tce_adam_once_*, what a sharded update runs
    behind its all-reduces)."""
    from tce_rl_amd.optim import FlatAdam
    ref = [torch.nn.Parameter(p.clone()) for p in _params(dtype, 0)]
    mine = [torch.nn.Parameter(p.clone().cuda()) for p in _params(dtype, 0)]
    o_ref = torch.optim.Adam(ref, lr=3e-3, weight_decay=wd)
    o_mine = FlatAdam(mine, lr=3e-3, weight_decay=wd)
    g = torch.Generator().manual_seed(1)
    for step in range(6):
        grads = [torch.randn(p.shape, generator=g, dtype=dtype) * (1 + step)
                 for p in ref]
        for p, q, gr in zip(ref, mine, grads):
            p.grad = gr.clone()
            if step % 2:                 # a fresh tensor, as autograd leaves it
                q.grad = gr.clone().cuda()
            else:                        # written into the bound flat view
                o_mine.bind_grads()
                q.grad.copy_(gr)
        before = torch.sqrt(sum((p.grad ** 2).sum() for p in ref))
        if clip > 0:                     # util_numerical.py:244-275
            torch.nn.utils.clip_grad_norm_(ref, clip)
        after = torch.sqrt(sum((p.grad ** 2).sum() for p in ref))
        o_ref.step()
        if once:
            norms = torch.zeros(2, dtype=dtype, device="cuda")
            o_mine.step_once(clip, norms_out=norms)
            nb, na = norms[0], norms[1]
            assert torch.equal(norms, o_mine.dev_state[1:3])
        else:
            nb, na = o_mine.step(clip)
        tol = 1e-5 if dtype == torch.float32 else 1e-12
        assert nb.item() == pytest.approx(before.item(), rel=tol)
        assert na.item() == pytest.approx(after.item(), rel=10 * tol)
        for p, q in zip(ref, mine):
            torch.testing.assert_close(q.detach().cpu(), p.detach(),
                                       rtol=20 * tol, atol=20 * tol)
    assert o_mine.host_step == 6 and o_mine.dev_state[0].item() == 6


def test_state_dict_is_interchangeable_with_torch_adam(tmp_path):
    from tce_rl_amd.optim import FlatAdam
    mine = [torch.nn.Parameter(p.cuda()) for p in _params(torch.float32, 2)]
    opt = FlatAdam(mine, lr=1e-3, weight_decay=1e-4)
    for p in mine:
        p.grad.copy_(torch.randn_like(p))
    opt.step()
    opt.step()
    path = tmp_path / "policy_optimizer_state_1"
    torch.save(opt.state_dict(), path)
    sd = torch.load(path, map_location="cpu")
    # torch's Adam accepts the file ...
    ref = [torch.nn.Parameter(p.detach().cpu().clone()) for p in mine]
    o_ref = torch.optim.Adam(ref, lr=1e-3, weight_decay=1e-4)
    o_ref.load_state_dict(sd)
    assert float(o_ref.state[ref[0]]["step"]) == 2
    # ... and a fresh FlatAdam continues from a torch-written one identically
    o_ref_sd = o_ref.state_dict()
    again = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    opt2 = FlatAdam(again, lr=5e-2)
    opt2.load_state_dict(o_ref_sd)
    assert opt2.param_groups[0]["lr"] == 1e-3 and opt2.host_step == 2
    g = [torch.randn(p.shape) for p in ref]
    for p, q, gr in zip(ref, again, g):
        p.grad = gr.clone()
        q.grad.copy_(gr)
    o_ref.step()
    opt2.step()
    for p, q in zip(ref, again):
        torch.testing.assert_close(q.detach().cpu(), p.detach(), rtol=2e-5,
                                   atol=2e-6)


def test_parameters_become_views_and_modules_keep_working():
    from tce_rl_amd.nn import MLP
    from tce_rl_amd.optim import FlatAdam
    torch.manual_seed(0)
    mlp = MLP("ValueFunction", 7, 1, [32, 32], "orthogonal", 1.0, "tanh", None,
              torch.float32, torch.device("cuda"))
    x = torch.randn(50, 7, device="cuda")
    y0 = mlp(x).detach().clone()
    opt = FlatAdam(list(mlp.parameters()), lr=1e-2)
    torch.testing.assert_close(mlp(x).detach(), y0)          # same values
    base = opt.flat_param.untyped_storage().data_ptr()
    assert all(p.untyped_storage().data_ptr() == base for p in mlp.parameters())
    opt.zero_grad()
    mlp(x).pow(2).mean().backward()                          # accumulates in place
    assert opt.flat_grad.abs().sum().item() > 0
    opt.step()
    assert (mlp(x).detach() - y0).abs().max().item() > 0


def test_fused_critic_adam_matches_separate_step():
    """tce_mlp_critic_f32 with the Adam step fused into the slab reduction ==
    gradient-only call followed by tce_adam_flat."""
    from tce_rl_amd import critic_ops
    from tce_rl_amd.nn import MLP
    from tce_rl_amd.optim import FlatAdam
    nets, opts = [], []
    for _ in range(2):
        torch.manual_seed(4)
        mlp = MLP("ValueFunction", 40, 1, [128, 128], "orthogonal", 1.0,
                  "relu", None, torch.float32, torch.device("cuda"))
        nets.append(mlp)
        opts.append(FlatAdam(list(mlp.parameters()), lr=2e-3,
                             weight_decay=1e-4))
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(33, 21, 40, device="cuda", generator=g)
    ret = torch.randn(33, 21, device="cuda", generator=g)
    runs = [critic_ops.EpochRunner(m, o.flat_grad) for m, o in zip(nets, opts)]
    for _ in range(4):
        s0 = runs[0].epoch(x, ret, ret, 0.0, adam=opts[0])
        s1 = runs[1].epoch(x, ret, ret, 0.0)
        opts[1].step(0.0, sumsq=s1[1:2])
        torch.testing.assert_close(s0, s1)
    for p, q in zip(nets[0].parameters(), nets[1].parameters()):
        torch.testing.assert_close(p.detach(), q.detach(), rtol=1e-6, atol=1e-7)
    assert opts[0].dev_state[0].item() == opts[1].dev_state[0].item() == 4


def test_cu_range_stream_runs_kernels():
    import ctypes
    from tce_rl_amd import _lib, ops
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.tce_stream_create_cu_range(0, 4, ctypes.byref(h)) == 0
    assert lib.tce_stream_create_cu_range(30, 4, ctypes.byref(ctypes.c_void_p())) != 0
    st = torch.cuda.ExternalStream(h.value)
    x = torch.randn(64, 100, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        stats = ops.moments(x)
    st.synchronize()
    assert stats is not None
    assert lib.tce_stream_destroy(h) == 0


def test_grad_scale_equals_scaling_the_gradient():
    """step(grad_scale=s) == step on s * grad (the multi-GPU mean of summed
    shard gradients), norms and clipping included."""
    from tce_rl_amd.optim import FlatAdam
    res = []
    for scaled in (True, False):
        ps = [torch.nn.Parameter(p.clone().cuda())
              for p in _params(torch.float32, 3)]
        opt = FlatAdam(ps, lr=1e-3, weight_decay=1e-3)
        g = torch.Generator().manual_seed(9)
        for _ in range(3):
            for p in ps:
                gr = torch.randn(p.shape, generator=g).cuda()
                p.grad.copy_(gr if scaled else gr * 0.125)
            nb, na = opt.step(0.3, grad_scale=0.125 if scaled else 1.0)
        res.append(([p.detach().clone() for p in ps], nb.item(), na.item()))
    for a, b in zip(res[0][0], res[1][0]):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-7)
    assert res[0][1] == pytest.approx(res[1][1], rel=1e-6)
    assert res[0][2] == pytest.approx(res[1][2], rel=1e-6)
