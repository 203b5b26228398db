"""torch.ops.tce_rl_amd.* (tce_rl_amd/torch_ops.py) against the direct op layer
(tce_rl_amd/ops.py, itself held to the oracle / golden vectors elsewhere) and
torch.library.opcheck (schema, fake tensors, autograd registration)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _chol(K, g, B=None):
    shape = (K, K) if B is None else (B, K, K)
    A = 0.2 * torch.randn(*shape, generator=g)
    return torch.linalg.cholesky(A @ A.transpose(-1, -2) + 0.5 * torch.eye(K)).cuda()


def test_rollout_ops_equal_the_op_layer():
    import tce_rl_amd.torch_ops  # noqa: F401
    from tce_rl_amd import ops
    ns = torch.ops.tce_rl_amd
    g = torch.Generator().manual_seed(0)
    N, T = 64, 100
    r = torch.randn(N, T, generator=g).cuda()
    v = torch.randn(N, T + 1, generator=g).cuda()
    d = torch.zeros(N, T, dtype=torch.bool, device="cuda")
    d[:, -1] = True
    tl = torch.zeros_like(d)
    adv, ret = ns.gae(r, v, d, tl, 0.99, 0.95, True)
    adv0, ret0 = ops.gae(r, v, d, tl, 0.99, 0.95, True)
    assert torch.equal(adv, adv0) and torch.equal(ret, ret0)
    pairs = torch.tensor([[0, 9], [10, 19], [20, 49], [50, 99]]).cuda()
    for mode in ("value_subtraction", "accumulate", "accumulated_rewards"):
        a = ns.segment_advantage(mode, r, v, adv, pairs, 0.99, True, 0.0)
        b = ops.segment_advantage(mode, r, v, adv, pairs, 0.99, True, 0.0)
        assert torch.equal(a, b), mode
    ev = torch.zeros(N, T, dtype=torch.bool, device="cuda")
    ev[::2, 40:] = True
    assert torch.equal(ns.mdp_reward(r, ev), ops.mdp_reward(r, ev))
    x = torch.randn(500, 12, generator=g).cuda()
    m1, v1 = torch.zeros(12, device="cuda"), torch.ones(12, device="cuda")
    m2, v2 = m1.clone(), v1.clone()
    ns.rms_update(x, m1, v1, 1e-4)
    ops.rms_update(x, m2, v2, 1e-4)
    assert torch.equal(m1, m2) and torch.equal(v1, v2)


@pytest.mark.parametrize("shared", [True, False])
def test_gaussian_ops_and_their_gradients(shared):
    import tce_rl_amd.torch_ops  # noqa: F401
    from tce_rl_amd import ops
    ns = torch.ops.tce_rl_amd
    g = torch.Generator().manual_seed(1)
    N, K = 96, 20
    x = torch.randn(N, K, generator=g).cuda()
    L = _chol(K, g) if shared else _chol(K, g, N)
    mean = torch.randn(N, K, generator=g).cuda()
    w = torch.randn(N, generator=g).cuda()
    # log-prob: value and gradients w.r.t. mean and L
    m1, L1 = mean.clone().requires_grad_(True), L.clone().requires_grad_(True)
    (ns.mvn_log_prob(x, m1, L1) * w).sum().backward()
    m2 = mean.clone().requires_grad_(True)
    L2 = (L if not shared else L).clone().requires_grad_(True)
    Lfull = L2 if not shared else ops.expand_shared(L2, N)
    (ops.mvn_log_prob(x, m2, Lfull) * w).sum().backward()
    torch.testing.assert_close(m1.grad, m2.grad, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(torch.tril(L1.grad), torch.tril(L2.grad),
                               rtol=1e-5, atol=1e-5)
    ref = torch.distributions.MultivariateNormal(
        mean.double(), scale_tril=(L.double() if not shared else
                                   L.double().expand(N, K, K))).log_prob(x.double())
    torch.testing.assert_close(ns.mvn_log_prob(x, mean, L).double(), ref,
                               rtol=1e-5, atol=1e-4)
    # maha and the mean projection
    y = torch.randn(N, K, generator=g).cuda()
    x1 = x.clone().requires_grad_(True)
    (ns.maha(x1, y, L) * w).sum().backward()
    x2 = x.clone().requires_grad_(True)
    (ops.maha(x2, y, L if not shared else ops.expand_shared(L, N)) * w).sum().backward()
    torch.testing.assert_close(x1.grad, x2.grad, rtol=1e-6, atol=1e-6)
    p1 = mean.clone().requires_grad_(True)
    out1 = ns.kl_mean_projection(p1, y, L, 0.05)
    (out1 * x).sum().backward()
    p2 = mean.clone().requires_grad_(True)
    out2 = ops.kl_mean_projection(p2, y, L if not shared else ops.expand_shared(L, N), 0.05)
    (out2 * x).sum().backward()
    assert torch.equal(out1, out2)
    torch.testing.assert_close(p1.grad, p2.grad, rtol=1e-6, atol=1e-6)


def test_kl_cov_projection_op():
    import tce_rl_amd.torch_ops  # noqa: F401
    from tce_rl_amd import ops
    g = torch.Generator().manual_seed(2)
    K = 24
    Lo = _chol(K, g)
    Ln = (Lo + 0.05 * torch.tril(torch.randn(K, K, generator=g)).cuda())[None]
    w = torch.randn(1, K, K, generator=g).cuda()
    a = Ln.clone().requires_grad_(True)
    proj, _ = torch.ops.tce_rl_amd.kl_cov_projection(a, Lo, 1e-3)
    (proj * w).sum().backward()
    b = Ln.clone().requires_grad_(True)
    proj0 = ops.kl_cov_projection(b, Lo, 1e-3)
    (proj0 * w).sum().backward()
    assert torch.equal(proj, proj0)
    torch.testing.assert_close(a.grad, b.grad, rtol=1e-6, atol=1e-7)


def test_critic_values_op():
    import tce_rl_amd.torch_ops  # noqa: F401
    from tce_rl_amd.nn import MLP
    torch.manual_seed(0)
    mlp = MLP("ValueFunction", 39, 1, [128, 128], "orthogonal", 1.0, "relu",
              None, torch.float32, torch.device("cuda"))
    x = torch.randn(3000, 48, device="cuda")[:, :39]       # strided rows
    ls = mlp.layers
    v = torch.ops.tce_rl_amd.critic_values(
        x, ls[0].weight, ls[0].bias, ls[1].weight, ls[1].bias, ls[2].weight,
        ls[2].bias, "relu")
    h = torch.relu(x.double() @ ls[0].weight.double().t() + ls[0].bias.double())
    h = torch.relu(h @ ls[1].weight.double().t() + ls[1].bias.double())
    ref = (h @ ls[2].weight.double().t() + ls[2].bias.double()).squeeze(-1)
    torch.testing.assert_close(v.double(), ref, rtol=2e-5, atol=2e-5)


def test_opcheck():
    import tce_rl_amd.torch_ops  # noqa: F401
    from torch.library import opcheck
    g = torch.Generator().manual_seed(3)
    ns = torch.ops.tce_rl_amd
    N, T, K = 16, 30, 6
    r = torch.randn(N, T, generator=g).cuda()
    v = torch.randn(N, T + 1, generator=g).cuda()
    d = torch.zeros(N, T, dtype=torch.bool, device="cuda")
    tests = ("test_schema", "test_faketensor", "test_autograd_registration")
    opcheck(ns.gae.default, (r, v, d, d, 0.99, 0.95, True), test_utils=tests)
    x = torch.randn(N, K, generator=g).cuda()
    L = _chol(K, g)
    m = torch.randn(N, K, generator=g).cuda().requires_grad_(True)
    opcheck(ns.mvn_log_prob.default, (x, m, L.clone().requires_grad_(True)),
            test_utils=tests)
    opcheck(ns.kl_mean_projection.default, (m, x, L, 0.05), test_utils=tests)
    mean, var = torch.zeros(K, device="cuda"), torch.ones(K, device="cuda")
    opcheck(ns.rms_update.default, (x, mean, var, 1e-4), test_utils=tests)
