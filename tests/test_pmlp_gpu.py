"""csrc/pmlp.hip: the policy mean nets the fused 128 x 2 float32 kernels do not
cover (float64 128 x 2: box pushing; 256 x 1 tanh: table tennis) against plain
PyTorch float64 references of MLP.forward (mprl/util/util_nn.py:225-246) and of
what autograd returns for it."""
import pytest
import torch

pytestmark = pytest.mark.gpu

ACTS = {"tanh": torch.tanh, "relu": torch.relu,
        "leaky_relu": torch.nn.functional.leaky_relu,
        "softplus": torch.nn.functional.softplus}
TOL = {torch.float32: (3e-5, 3e-6), torch.float64: (1e-11, 1e-12)}


def make_mlp(din, hidden, dout, act, dtype, seed=0, gain=1.0):
    from tce_rl_amd.nn import MLP
    torch.manual_seed(seed)
    net = MLP("t", din, dout, hidden, "orthogonal", gain, act, None, dtype,
              torch.device("cuda"))
    with torch.no_grad():                   # biases start at zero: move them
        for p in net.parameters():
            if p.dim() == 1:
                p.copy_(0.1 * torch.randn(p.shape, dtype=dtype))
    return net


def ref_forward(params, x, act):
    f = ACTS[act]
    h = x
    for i in range(0, len(params) - 2, 2):
        h = f(h @ params[i].T + params[i + 1])
    return h @ params[-2].T + params[-1]


SHAPES = [  # (din, hidden, dout)
    (22, [128, 128], 63),     # box pushing
    (22, [256], 28),          # table tennis
    (40, [128, 128], 24),     # metaworld
    (1, [128], 1), (33, [128], 64), (64, [128, 128], 17), (7, [256], 64)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("din,hidden,dout", SHAPES)
@pytest.mark.parametrize("act", list(ACTS))
@pytest.mark.parametrize("N", [1, 33, 1000])
def test_forward_backward(dtype, din, hidden, dout, act, N):
    from tce_rl_amd import pmlp_ops
    net = make_mlp(din, hidden, dout, act, dtype)
    assert pmlp_ops.supported(net)
    g = torch.Generator().manual_seed(N + din)
    xd = torch.randn(N, din + 3, generator=g, dtype=dtype).cuda()
    x = xd[:, :din]                                         # strided rows
    keep = {}
    out = pmlp_ops.forward(net, x, keep=keep)
    ref_p = [p.detach().cpu().double().requires_grad_() for p in net.parameters()]
    ref = ref_forward(ref_p, x.cpu().double(), act)
    rtol, atol = TOL[dtype]
    scale = max(1.0, float(ref.abs().max()))
    torch.testing.assert_close(out.cpu().double(), ref.detach(), rtol=rtol,
                               atol=atol * scale)
    # hidden activations kept for the backward == the forward-only launch
    out2 = pmlp_ops.forward(net, x)
    assert torch.equal(out, out2)
    go = torch.randn(N, dout, generator=g, dtype=dtype)
    grad = pmlp_ops.backward(net, keep, go.cuda())
    (ref * go.double()).sum().backward()
    ref_g = torch.cat([p.grad.reshape(-1) for p in ref_p])
    gs = max(1.0, float(ref_g.abs().max()))
    torch.testing.assert_close(grad.cpu().double(), ref_g, rtol=10 * rtol,
                               atol=10 * atol * gs)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("budget", [0, 32])
def test_full_size_rows_and_repeatability(dtype, budget):
    """8192 rows (box pushing), all tiles / slabs in play; a second launch gives
    the same bits; a CU budget (beside the critic) changes the slab count only."""
    from tce_rl_amd import pmlp_ops
    from tce_rl_amd._lib import call
    net = make_mlp(22, [128, 128], 63, "leaky_relu", dtype)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(8192, 22, generator=g, dtype=dtype).cuda()
    go = (torch.randn(8192, 63, generator=g, dtype=dtype) / 8192).cuda()
    call("tce_set_cu_budget", budget)
    try:
        keep = {}
        out = pmlp_ops.forward(net, x, keep=keep)
        grad = pmlp_ops.backward(net, keep, go)
        grad2 = pmlp_ops.backward(net, keep, go)
    finally:
        call("tce_set_cu_budget", 0)
    assert torch.equal(grad, grad2)
    ref_p = [p.detach().cpu().double().requires_grad_() for p in net.parameters()]
    ref = ref_forward(ref_p, x.cpu().double(), "leaky_relu")
    rtol, atol = TOL[dtype]
    torch.testing.assert_close(out.cpu().double(), ref.detach(), rtol=rtol,
                               atol=atol * float(ref.abs().max()))
    (ref * go.cpu().double()).sum().backward()
    ref_g = torch.cat([p.grad.reshape(-1) for p in ref_p])
    torch.testing.assert_close(grad.cpu().double(), ref_g, rtol=10 * rtol,
                               atol=10 * atol * float(ref_g.abs().max()))


def test_unsupported_shapes_are_refused():
    from tce_rl_amd import _lib
    lib = _lib.load()
    assert lib.tce_pmlp_supported(22, 128, 2, 63, 8)
    assert lib.tce_pmlp_supported(22, 256, 1, 28, 4)
    assert not lib.tce_pmlp_supported(22, 256, 2, 28, 8)
    assert not lib.tce_pmlp_supported(65, 128, 2, 28, 4)
    assert not lib.tce_pmlp_supported(22, 128, 2, 65, 4)
    assert not lib.tce_pmlp_supported(22, 96, 2, 28, 4)
    x = torch.zeros(4, 22, device="cuda")
    rc = lib.tce_pmlp_forward_f32(x.data_ptr(), 22, 4, 22, 96, 2, 28, 0,
                                  x.data_ptr(), None, None, x.data_ptr(), None)
    assert rc != 0 and b"tce_pmlp_supported" in lib.tce_last_error()
