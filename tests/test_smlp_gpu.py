"""csrc/smlp.hip: the row kernels for the black-box agent's small networks
(D_in -> H -> H -> D_out, H in {32, 64}) against plain PyTorch references of
the same operations: forward, critic epochs (value loss + backward + clip +
Adam, mprl/rl/agent/black_box_agent.py:105-157) and policy epochs
(black_box_agent.py:159-389) against the CPU oracle's pieces."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ACTS = {"tanh": torch.tanh, "relu": torch.relu,
        "leaky_relu": torch.nn.functional.leaky_relu,
        "softplus": torch.nn.functional.softplus}


def make_mlp(din, H, dout, act, gain=1.0, seed=0):
    from tce_rl_amd.nn import MLP
    torch.manual_seed(seed)
    return MLP("t", din, dout, [H, H], "orthogonal", gain, act, None,
               torch.float32, torch.device("cuda"))


def ref_forward(params, x, act):
    w1, b1, w2, b2, w3, b3 = params
    f = ACTS[act]
    h = f(x @ w1.T + b1)
    h = f(h @ w2.T + b2)
    return h @ w3.T + b3


@pytest.mark.parametrize("din,H,dout", [(39, 32, 1), (39, 32, 20), (1, 32, 3),
                                        (17, 64, 33), (64, 64, 64),
                                        (22, 32, 63)])
@pytest.mark.parametrize("act", list(ACTS))
@pytest.mark.parametrize("N", [1, 63, 200, 4101])
def test_forward(din, H, dout, act, N):
    from tce_rl_amd import smlp_ops
    net = make_mlp(din, H, dout, act)
    if not smlp_ops.supported(net):
        pytest.skip("does not fit the LDS")
    g = torch.Generator().manual_seed(N)
    x = torch.randn(N, din + 3, generator=g)[:, :din]      # strided rows
    xd = torch.randn(N, din + 3, generator=g).cuda()
    xd[:, :din] = x.cuda()
    out = smlp_ops.forward(net, xd[:, :din])
    ref = ref_forward([p.detach().cpu().double() for p in net.parameters()],
                      x.double(), act)
    torch.testing.assert_close(out.cpu().double(), ref, rtol=2e-5, atol=2e-6)


class _FakeAgent:
    """What smlp_ops.critic_update / policy_update read from the agent."""

    def __init__(self, **kw):
        from tce_rl_amd.dist import DistContext
        self.dist = DistContext()
        self.num_minibatchs = 1
        self.clip_critic = self.clip_grad_norm = 0.0
        self.entropy_penalty_coef = 0.0
        self.set_variance = False
        self._policy_group = None
        self.__dict__.update(kw)


@pytest.mark.parametrize("din,H,act,clip,clip_gn,wd,N", [
    (39, 32, "relu", 0.0, 0.0, 0.0, 4096), (39, 32, "relu", 0.3, 0.5, 0.0, 300),
    (11, 64, "tanh", 0.0, 0.0, 1e-3, 129), (64, 32, "leaky_relu", 0.2, 0.0, 0.0, 64),
    (5, 64, "softplus", 0.0, 2.0, 0.0, 1000), (39, 32, "relu", 0.0, 0.0, 0.0, 70000)])
def test_critic_epochs_match_torch_adam(din, H, act, clip, clip_gn, wd, N):
    """E epochs of the ONE-launch critic epoch == E steps of torch autograd +
    clip_grad_norm_ + torch.optim.Adam in float64 on the same data."""
    from tce_rl_amd import smlp_ops
    from tce_rl_amd.optim import FlatAdam
    from tce_rl_amd.rl.critic import ValueFunction
    E = 4
    net = make_mlp(din, H, 1, act, seed=3)
    ref = [p.detach().cpu().double().clone().requires_grad_(True)
           for p in net.parameters()]
    opt = FlatAdam(list(net.parameters()), lr=3e-3, weight_decay=wd)
    ropt = torch.optim.Adam(ref, lr=3e-3, weight_decay=wd)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, din, generator=g)
    ret = 2.0 * torch.randn(N, generator=g)
    old = ret + 0.5 * torch.randn(N, generator=g)
    critic = ValueFunction.__new__(ValueFunction)
    critic.net = net
    agent = _FakeAgent(critic=critic, critic_optimizer=opt, epochs_critic=E,
                       clip_critic=clip, clip_grad_norm=clip_gn)
    assert smlp_ops.critic_supported(agent)
    rec = smlp_ops.critic_update(agent, x.cuda(), ret.cuda(), old.cuda())
    rows = []
    for _ in range(E):
        v = ref_forward(ref, x.double(), act).squeeze(-1)
        l = (ret.double() - v) ** 2
        if clip > 0:
            vc = old.double() + (v - old.double()).clamp(-clip, clip)
            l = torch.max(l, (vc - ret.double()) ** 2)
        loss = l.mean()
        ropt.zero_grad()
        loss.backward()
        before = torch.sqrt(sum((p.grad ** 2).sum() for p in ref)).item()
        if clip_gn > 0:
            torch.nn.utils.clip_grad_norm_(ref, clip_gn)
        after = torch.sqrt(sum((p.grad ** 2).sum() for p in ref)).item()
        ropt.step()
        rows.append([loss.item(), before, after])
    np.testing.assert_allclose(rec.cpu().numpy(), np.array(rows), rtol=3e-4,
                               atol=1e-6)
    for p, q in zip(net.parameters(), ref):
        torch.testing.assert_close(p.detach().cpu().double(), q.detach(),
                                   rtol=0, atol=3e-5)
    assert opt.host_step == E and float(opt.dev_state[0]) == E


def test_critic_epochs_are_deterministic():
    """Fixed slab order: two runs end bit-identical (many workgroups)."""
    from tce_rl_amd import smlp_ops
    from tce_rl_amd.optim import FlatAdam
    from tce_rl_amd.rl.critic import ValueFunction
    outs = []
    for _ in range(2):
        net = make_mlp(39, 32, 1, "relu", seed=5)
        opt = FlatAdam(list(net.parameters()), lr=1e-3)
        g = torch.Generator().manual_seed(2)
        x = torch.randn(20000, 39, generator=g).cuda()
        ret = torch.randn(20000, generator=g).cuda()
        critic = ValueFunction.__new__(ValueFunction)
        critic.net = net
        agent = _FakeAgent(critic=critic, critic_optimizer=opt,
                           epochs_critic=5)
        smlp_ops.critic_update(agent, x, ret, ret)
        outs.append(opt.flat_param.clone())
    assert torch.equal(*outs)


@pytest.fixture
def general_finish():
    """tce_bb_finish_general(1) for the test, the default back afterwards."""
    from tce_rl_amd import _lib
    lib = _lib.load()
    lib.tce_bb_finish_general(1)
    yield
    lib.tce_bb_finish_general(0)


@pytest.mark.parametrize("std_only,K,H,din,N,ent,set_var", [
    (True, 20, 32, 39, 300, 0.0, True), (False, 12, 32, 39, 130, 0.01, False),
    (False, 12, 64, 16, 64, 0.0, False), (True, 6, 32, 9, 4096, 0.0, False),
    (False, 33, 32, 39, 77, 0.0, False),
    # more than 1024 tiles: workgroups walk several tiles (tile DMA inside the
    # loop, slabs accumulated)
    (True, 6, 32, 9, 66000, 0.0, True)])
@pytest.mark.parametrize("diag_kernels", [True, False, "general_finish"])
def test_policy_epochs_match_the_oracle_pieces(std_only, K, H, din, N, ent,
                                               set_var, diag_kernels,
                                               monkeypatch, request):
    """E policy epochs on the row kernels == E epochs built from the CPU
    oracle's pieces (oracle/kl_oracle.py project / trust_region_loss,
    oracle/tce_oracle.py mvn_log_prob / surrogate_loss) with torch autograd
    and torch.optim.Adam, float64."""
    from oracle import kl_oracle as KO
    from oracle import tce_oracle as O
    from tce_rl_amd import ops, smlp_ops
    from tce_rl_amd.optim import FlatAdam
    from tce_rl_amd.rl.policy import BlackBoxPolicy
    from tce_rl_amd.rl import projection_factory
    E, act = 3, "relu"
    if diag_kernels == "general_finish":
        # the diagonal row kernels with the general finish kernel behind them
        if not std_only:
            pytest.skip("full factors always take the general finish kernel")
        request.getfixturevalue("general_finish")
    if not diag_kernels:
        if not std_only:
            pytest.skip("full factors always take the K x K kernels")
        monkeypatch.setenv("TCE_BB_DIAG", "0")    # std_only on the general kernels
    torch.manual_seed(4)
    pol = BlackBoxPolicy(
        dim_in=din, dim_out=K,
        mean_net_args=dict(avg_neuron=H, num_hidden=2, shape=0.0),
        variance_net_args=dict(std_only=std_only, contextual=False),
        init_method="orthogonal", out_layer_gain=0.3, act_func_hidden=act,
        act_func_last=None, dtype="float32", device="cuda", min_std=1e-4)
    proj = projection_factory(
        "KLProjectionLayer", proj_type="kl", mean_bound=0.02, cov_bound=0.002,
        trust_region_coeff=5.0, entropy_schedule=False, action_dim=K,
        total_train_steps=100, dtype="float32", device="cuda")
    params = list(pol.mean_net.parameters()) + [pol.variance_net.variable]
    opt = FlatAdam(params, lr=3e-3)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, din, generator=g)
    with torch.no_grad():
        # the old distribution: a perturbed copy of the new one
        mean_old = (pol.mean_net(x.cuda()).cpu()
                    + 0.05 * torch.randn(N, K, generator=g))
        v_old = pol.variance_net.variable.detach().cpu() \
            + 0.1 * torch.randn(pol.variance_net.variable.numel(), generator=g)
        L_old = O.vector_to_cholesky(v_old[None], K, 1e-4, std_only)[0]
        actions = mean_old + (L_old @ torch.randn(N, K, 1, generator=g))[..., 0]
        lp_old = O.mvn_log_prob(actions, mean_old,
                                L_old.expand(N, -1, -1))
        adv = torch.randn(N, generator=g)
    ref = [p.detach().cpu().double().clone().requires_grad_(True)
           for p in params]
    ropt = torch.optim.Adam(ref, lr=3e-3)
    agent = _FakeAgent(policy=pol, projection=proj, policy_optimizer=opt,
                       epochs_policy=E, entropy_penalty_coef=ent,
                       set_variance=set_var)
    L_old_d = ops.expand_shared(L_old.cuda(), N)
    assert smlp_ops.policy_supported(agent, L_old_d)
    rec, mean_new, L_new, pmean, pL = smlp_ops.policy_update(
        agent, x.cuda(), actions.cuda(), lp_old.cuda(), adv.cuda(),
        mean_old.cuda(), L_old_d, None)
    d = lambda t: t.double()
    rows = []
    no_beta = torch.tensor(-float("inf"), dtype=torch.float64)
    for _ in range(E):
        mean = ref_forward(ref[:6], d(x), act)
        L = O.vector_to_cholesky(ref[6][None], K, 1e-4, std_only) \
            .expand(N, -1, -1)
        pm, pLr = KO.project(mean, L, d(mean_old), d(L_old).expand(N, -1, -1),
                             0.02, 0.002, no_beta, contextual_std=False)
        lp = O.mvn_log_prob(d(actions), pm, pLr)
        s_loss, _ = O.surrogate_loss(d(adv), lp, d(lp_old))
        entropy = KO.entropy(pLr).mean()
        e_loss = -ent * entropy
        tr = KO.trust_region_loss(mean, L, pm, pLr, 5.0, not set_var)
        total = s_loss + e_loss + tr
        ropt.zero_grad()
        total.backward()
        gn = torch.sqrt(sum((p.grad ** 2).sum() for p in ref)).item()
        ropt.step()
        with torch.no_grad():      # kl_old_new_proj, black_box_agent.py:391-436
            Lo = d(L_old).expand(N, -1, -1)
            kl = [t.mean().item()
                  for p, q in (((mean, L), (d(mean_old), Lo)), ((mean, L), (pm, pLr)),
                               ((pm, pLr), (d(mean_old), Lo)))
                  for t in KO.gaussian_kl_details(p[0], p[1], q[0], q[1])]
        rows.append([s_loss.item(), e_loss.item(), tr.item(), total.item(),
                     entropy.item(), gn, gn] + kl)
        last = (mean.detach(), L[0].detach(), pm.detach(), pLr[0].detach())
    got = rec.cpu().numpy()
    assert got.shape == (E, 19)
    np.testing.assert_allclose(got[:, :7], np.array(rows)[:, :7], rtol=2e-3,
                               atol=2e-5)
    # the 12 KL means: float32 differences of O(K) terms against a float64 reference
    np.testing.assert_allclose(got[:, 7:], np.array(rows)[:, 7:], rtol=2e-3,
                               atol=3e-5)
    for p, q in zip(params, ref):
        torch.testing.assert_close(p.detach().cpu().double(), q.detach(),
                                   rtol=0, atol=5e-5)
    torch.testing.assert_close(mean_new.cpu().double(), last[0], rtol=0,
                               atol=1e-4)
    torch.testing.assert_close(pmean.cpu().double(), last[2], rtol=0,
                               atol=1e-4)
    torch.testing.assert_close(L_new.cpu().double(), last[1], rtol=1e-5,
                               atol=1e-6)
    torch.testing.assert_close(pL.cpu().double(), last[3], rtol=1e-4,
                               atol=1e-6)


@pytest.mark.parametrize("N,din,dout,transposed", [
    (4096, 128, 24, True), (4096, 24, 128, False), (70, 128, 63, True),
    (1000, 63, 128, False), (129, 256, 36, True), (5, 7, 3, True),
    (300, 20, 100, False), (70000, 40, 24, True)])
def test_lin_rows(N, din, dout, transposed):
    """tce_lin_rows_f32 (the output layer of the policy mean net and its input
    gradient) against torch in float64."""
    from tce_rl_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(N + din)
    x = torch.randn(N, din + 4, generator=g).cuda()[:, :din]
    W = torch.randn(*((dout, din) if transposed else (din, dout)),
                    generator=g).cuda()
    b = torch.randn(dout, generator=g).cuda() if transposed else None
    y = torch.empty(N, dout, device="cuda")
    call("tce_lin_rows_f32", ptr(x), x.stride(0), N, din, dout, ptr(W),
         int(transposed), ptr(b), ptr(y), stream())
    A = W.double().t() if transposed else W.double()
    ref = x.double() @ A + (b.double() if b is not None else 0)
    torch.testing.assert_close(y.double(), ref, rtol=1e-5, atol=1e-4)
