"""GPU parity of the ProDMP kernels (time grid, parameter sampling,
trajectory, pair-wise log-prob forward + backward) against the CPU oracle and
the golden fixture generated through the reference's own log_prob code."""
import numpy as np
import pytest
import torch

from oracle import tce_oracle as O
from oracle.prodmp_oracle import ProDMPOracle, pair_log_prob

pytestmark = pytest.mark.gpu
T_ = torch.as_tensor

CFGS = {
    "metaworld": dict(num_dof=4, num_basis=8, tau=5, alpha_phase=3, alpha=10,
                      dt=0.0125, basis_bandwidth_factor=5, weights_scale=0.1,
                      goal_scale=0.1, relative_goal=True),
    "metaworld_nb5": dict(num_dof=4, num_basis=5, tau=5, alpha_phase=3,
                          alpha=10, dt=0.0125, basis_bandwidth_factor=5,
                          weights_scale=0.1, goal_scale=0.1,
                          relative_goal=True),
    "box_push": dict(num_dof=7, num_basis=8, tau=2.0, alpha_phase=3, alpha=10,
                     dt=0.02, basis_bandwidth_factor=3, weights_scale=0.3,
                     goal_scale=0.3, relative_goal=False),
    "table_tennis": dict(num_dof=7, num_basis=3, tau=0.75, delay=0.3,
                         alpha_phase=3, alpha=25, dt=0.008,
                         basis_bandwidth_factor=3, weights_scale=0.7,
                         goal_scale=0.1, relative_goal=True),
}
HORIZON = {"metaworld": 500, "metaworld_nb5": 500, "box_push": 100,
           "table_tennis": 350}


@pytest.fixture(scope="module")
def ops():
    from tce_rl_amd import ops
    return ops


def make(name, dtype):
    from tce_rl_amd.mp import ProDMP
    return ProDMP(dtype=dtype, device="cuda", **CFGS[name]), \
        ProDMPOracle(dtype=dtype, **CFGS[name])


def affine(times_cpu):
    """The oracle's own time grid on the GPU, tagged like ops.times() output:
    isolates the kernels under test from the last-bit differences of the
    float32 linspace weights (machine dependent on the CPU side)."""
    t = times_cpu.cuda()
    t._tce_affine = True
    return t


def inputs(name, N, dtype, seed=0, uniform_t0=True):
    cfg = CFGS[name]
    dof, K = cfg["num_dof"], cfg["num_dof"] * (cfg["num_basis"] + 1)
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g, dtype=dtype)
    mean = 0.5 * rn(N, K)
    L = O.vector_to_cholesky(
        torch.cat([rn(N, K), 0.05 * rn(N, K * (K - 1) // 2)], -1), K, 1e-4,
        False)
    eps = rn(N, K)
    y0 = torch.rand(N, dof, generator=g, dtype=dtype) * 2 - 1
    v0 = 0.1 * rn(N, dof)
    t0 = torch.zeros(N, dtype=dtype) if uniform_t0 else \
        torch.rand(N, generator=g, dtype=dtype) * 0.2
    return mean, L, eps, t0, y0, v0


def test_times_golden(ops, golden):
    """Same two-sided linspace formula as torch; the CPU reference evaluates it
    in 8-lane vector chunks (base + i*step), so the last bit may differ."""
    g = golden("times")
    for i in range(3):
        out = ops.times(T_(g[f"t0_{i}"]).cuda(), float(g[f"dt_{i}"]),
                        int(g[f"T_{i}"]))
        np.testing.assert_allclose(out.cpu().numpy(), g[f"times_{i}"],
                                   rtol=3e-7, atol=1e-7)


@pytest.mark.parametrize("name", list(CFGS))
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("uniform_t0", [True, False])
def test_traj_vs_oracle(ops, name, dtype, uniform_t0):
    mp, oracle = make(name, dtype)
    N, T = 37, HORIZON[name]
    mean, L, eps, t0, y0, v0 = inputs(name, N, dtype, 1, uniform_t0)
    times_cpu = O.get_times(t0, CFGS[name]["dt"], T)
    pos, vel = oracle.sample_trajectories(times_cpu, mean, L, t0, y0, v0, eps)
    ref = torch.cat([pos, vel], -1)
    tg = ops.times(t0.cuda(), CFGS[name]["dt"], T)
    torch.testing.assert_close(tg.cpu(), times_cpu, rtol=3e-7, atol=2e-7)
    tg = affine(times_cpu)
    w = ops.mvn_rsample(mean.cuda(), L.cuda(), eps.cuda())
    wtol = 1e-5 if dtype == torch.float32 else 1e-12
    torch.testing.assert_close(w.cpu(), O.mvn_rsample(mean, L, eps),
                               rtol=wtol, atol=wtol)
    out = ops.prodmp_traj(mp, tg, w, t0.cuda(), y0.cuda(), v0.cuda())
    tol = 2e-5 if dtype == torch.float32 else 1e-10
    torch.testing.assert_close(out.cpu(), ref, rtol=tol, atol=tol)
    # an untagged (arbitrary) times tensor takes the per-element basis path
    out2 = ops.prodmp_traj(mp, tg.clone(), w, t0.cuda(), y0.cuda(), v0.cuda())
    torch.testing.assert_close(out2.cpu(), ref, rtol=tol, atol=tol)


def test_traj_shared_L_and_boundary_conditions(ops):
    """Self-checks that stand in for mp_pytorch goldens: y(t0)=y0, dy(t0)=v0,
    vel = d pos/dt, convergence to the goal."""
    name, dtype = "metaworld", torch.float64
    mp, oracle = make(name, dtype)
    N, T, dt = 8, 500, CFGS[name]["dt"]
    mean, L, eps, t0, y0, v0 = inputs(name, N, dtype, 2)
    Ls = ops.expand_shared(L[0].cuda(), N)
    w = ops.mvn_rsample(mean.cuda(), Ls, eps.cuda())
    torch.testing.assert_close(
        w.cpu(), O.mvn_rsample(mean, L[:1].expand(N, -1, -1), eps))
    # times starting AT t0: first sample must reproduce the initial condition
    tt = (t0[:, None] + dt * torch.arange(T, dtype=dtype)[None, :]).cuda()
    out = ops.prodmp_traj(mp, tt, w, t0.cuda(), y0.cuda(), v0.cuda()).cpu()
    D = CFGS[name]["num_dof"]
    torch.testing.assert_close(out[:, 0, :D], y0, rtol=1e-9, atol=1e-9)
    torch.testing.assert_close(out[:, 0, D:], v0, rtol=1e-9, atol=1e-9)
    fd = (out[:, 2:, :D] - out[:, :-2, :D]) / (2 * dt)
    assert (fd - out[:, 1:-1, D:]).abs().max() < 2e-3
    tl = torch.full((N, 1), 24.9, dtype=dtype).cuda()
    end = ops.prodmp_traj(mp, tl, w, t0.cuda(), y0.cuda(), v0.cuda()).cpu()
    goal = mp.scale[-1] * w.cpu().reshape(N, D, -1)[..., -1] + y0
    assert (end[:, 0, :D] - goal).abs().max() < 1e-4


@pytest.mark.parametrize("name", list(CFGS))
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_kernel_trajectory_solves_the_dmp_ode(ops, name, dtype):
    """The trajectory KERNEL (table built by tce_rl_amd/mp/prodmp.py,
    interpolated and contracted on the GPU) against a scipy integration of the
    DMP ODE written from the paper (tests/prodmp_ode.py; shares no code with the
    oracle or the product).  Bounds as in tests/test_prodmp_ode_cpu.py: the
    tables' own second-order error, u = (alpha dt / tau)^2."""
    from prodmp_ode import DMPODE
    from tce_rl_amd.mp import ProDMP
    cfg = CFGS[name]
    mp = ProDMP(dtype=dtype, device="cuda", **cfg)
    ode = DMPODE(**{k: v for k, v in cfg.items() if k != "num_dof"})
    N, T, dt, dof = 3, HORIZON[name], cfg["dt"], cfg["num_dof"]
    K = dof * (cfg["num_basis"] + 1)
    g = np.random.default_rng(5)
    w = g.normal(size=(N, K))
    y0 = g.uniform(-1, 1, size=(N, dof))
    v0 = 0.3 * g.normal(size=(N, dof))
    times = dt * np.arange(1, T + 1)
    dev = lambda a: torch.as_tensor(a, dtype=dtype).cuda()
    t0 = torch.zeros(N, dtype=dtype).cuda()
    out = ops.prodmp_traj(mp, dev(np.tile(times, (N, 1))), dev(w), t0,
                          dev(y0), dev(v0)).double().cpu().numpy()
    u = (cfg["alpha"] * dt / cfg["tau"]) ** 2
    fp = 0.0 if dtype == torch.float64 else 2e-5
    for n in range(N):
        p_ref, v_ref = ode.trajectory(times, w[n], 0.0, y0[n], v0[n])
        assert np.abs(out[n, :, :dof] - p_ref).max() <= \
            (0.25 * u + fp) * np.abs(p_ref).max()
        assert np.abs(out[n, :, dof:] - v_ref).max() <= \
            (0.5 * u + fp) * np.abs(v_ref).max()


def test_pair_logprob_golden_plumbing(ops, golden):
    """Fixture produced by the reference's TemporalCorrelatedPolicy.log_prob."""
    from tce_rl_amd.mp import ProDMP
    g = golden("pair_logprob_plumbing")
    for tag in ("mw", "bp"):
        cfg = {k[len(tag) + 5:]: g[k].item() for k in g.files
               if k.startswith(tag + "_cfg_")}
        mp = ProDMP(dtype=torch.float32, device="cuda", **cfg)
        a = lambda k: T_(g[f"{tag}_{k}"]).cuda()
        times = affine(T_(g[f"{tag}_times"]))
        w = ops.mvn_rsample(a("mean"), a("L"), a("eps"))
        traj = ops.prodmp_traj(mp, times, w, a("t0"), a("y0"), a("v0"))
        np.testing.assert_allclose(traj.cpu().numpy(), g[f"{tag}_traj"],
                                   rtol=2e-5, atol=2e-5)
        lp = ops.pair_log_prob(mp, a("traj"), a("mean"), a("L"), times,
                               a("t0"), a("y0"), a("v0"), a("pairs"))
        # north_star: 1e-5 on log-probs.  The float32 likelihood is
        # conditioning-limited (C = H Sigma H^T + 1e-4 I is nearly singular):
        # the reference's OWN float32 output sits 3 - 5e-6 of max |logp| away
        # from the float64 value on this fixture, so the bar is set against the
        # float64 truth: (1) |hip_f32 - truth| <= LOGP_REL of max |truth|
        # (seen 5.3e-7 since the kernels form and factor C in double, round
        # 6; LOGP_REL = 3e-6 keeps a regression visible, north_star's bound
        # is 1e-5), (2) no further from the truth than the reference is,
        # (3) within that distance of the reference's float32 output.
        ref32 = g[f"{tag}_logp"]
        o64 = ProDMPOracle(dtype=torch.float64, **cfg)
        d = lambda k: T_(g[f"{tag}_{k}"]).double()
        truth = pair_log_prob(o64, d("traj"), d("mean"), d("L"), d("times"),
                              d("t0"), d("y0"), d("v0"),
                              T_(g[f"{tag}_pairs"])).numpy()
        scale = np.abs(truth).max()
        err_ours = np.abs(lp.cpu().numpy() - truth).max()
        err_ref = np.abs(ref32 - truth).max()
        assert err_ours <= LOGP_REL * scale, (err_ours, scale)
        assert err_ours <= err_ref, (err_ours, err_ref)
        assert np.abs(lp.cpu().numpy() - ref32).max() <= \
            err_ref + LOGP_REL * scale


# |float32 kernel - float64 truth| <= LOGP_REL * max |truth| (north_star: 1e-5;
# largest seen over every case of this file 9.8e-7, scripts/probe_logp_tol.py)
LOGP_REL = 3e-6
NORTH_STAR_LOGP_REL = 1e-5
assert LOGP_REL <= NORTH_STAR_LOGP_REL


_PL_NAMES = ["metaworld", "metaworld_nb5", "box_push", "table_tennis"]
# (name, shared, uniform_t0, N, form): N 300 = the shared-L fast path (N >= 256),
# whose two kernel forms -- "static": registers, built for the shipped (dof,
# basis count) shapes; "general": runtime shapes -- differ there only
_PL_CASES = [(n, sh, u, 6, "static") for n in _PL_NAMES for sh in (False, True)
             for u in (True, False)] + \
    [(n, True, u, 300, "static") for n in _PL_NAMES for u in (True, False)] + \
    [(n, True, True, 300, "general") for n in _PL_NAMES]


@pytest.mark.parametrize("name,shared,uniform_t0,N,form", _PL_CASES)
def test_pair_logprob_fwd_bwd_vs_oracle(ops, name, shared, uniform_t0, N, form):
    from tce_rl_amd._lib import call
    call("tce_pair_env_static", int(form == "static"))
    try:
        _pair_logprob_fwd_bwd_vs_oracle(ops, name, shared, uniform_t0, N)
    finally:
        call("tce_pair_env_static", 1)


def _pair_logprob_fwd_bwd_vs_oracle(ops, name, shared, uniform_t0, N):
    dtype = torch.float64
    mp, oracle = make(name, dtype)
    T = HORIZON[name]
    mean, L, eps, t0, y0, v0 = inputs(name, N, dtype, 3, uniform_t0)
    if shared:
        L = L[:1].expand(N, -1, -1).contiguous()
    times_cpu = O.get_times(t0, CFGS[name]["dt"], T)
    pos, vel = oracle.sample_trajectories(times_cpu, mean, L, t0, y0, v0, eps)
    traj = torch.cat([pos, vel], -1)
    torch.manual_seed(1)
    pairs = O.get_time_pairs(T, dict(num_select=25, fixed_interval=True))
    wgt = torch.randn(N, pairs.shape[0], dtype=dtype)
    # oracle forward / backward (torch autograd through the restatement)
    m_c = mean.clone().requires_grad_(True)
    if shared:
        Lb_c = L[0].clone().requires_grad_(True)
        L_c = Lb_c[None].expand(N, -1, -1)
    else:
        Lb_c = L.clone().requires_grad_(True)
        L_c = Lb_c
    lp_ref = pair_log_prob(oracle, traj, m_c, L_c, times_cpu, t0, y0, v0, pairs)
    (lp_ref * wgt).sum().backward()
    # HIP forward / backward
    m_g = mean.cuda().requires_grad_(True)
    if shared:
        Lb_g = L[0].cuda().requires_grad_(True)
        L_g = ops.expand_shared(Lb_g, N)
    else:
        Lb_g = L.cuda().requires_grad_(True)
        L_g = Lb_g
    tg = affine(times_cpu)
    lp = ops.pair_log_prob(mp, traj.cuda(), m_g, L_g, tg, t0.cuda(), y0.cuda(),
                           v0.cuda(), pairs.cuda())
    torch.testing.assert_close(lp.cpu(), lp_ref.detach(), rtol=1e-8, atol=1e-8)
    (lp * wgt.cuda()).sum().backward()
    torch.testing.assert_close(m_g.grad.cpu(), m_c.grad, rtol=1e-7, atol=1e-7)
    gl_ref = torch.tril(Lb_c.grad)
    torch.testing.assert_close(torch.tril(Lb_g.grad.cpu()), gl_ref, rtol=1e-7,
                               atol=1e-7)


@pytest.mark.parametrize("name,shared,uniform_t0,N,form", _PL_CASES + [
    (n, False, True, 300, "static") for n in _PL_NAMES])
def test_pair_logprob_float32_within_north_star_of_the_float64_truth(
        ops, name, shared, uniform_t0, N, form):
    """Every kernel form (per-env covariance, shared factor at 6 envs on the
    general kernel, the fast path's register / run-time-shape kernels, equal
    and differing init times) in FLOAT32 against the float64 oracle on the same
    float32 inputs: north_star's 1e-5 of max |logp| (asserted at LOGP_REL)."""
    from tce_rl_amd._lib import call
    dtype = torch.float32
    cfg = CFGS[name]
    mp = make(name, dtype)[0]
    o64 = ProDMPOracle(dtype=torch.float64, **cfg)
    T = HORIZON[name]
    mean, L, eps, t0, y0, v0 = inputs(name, N, dtype, 3, uniform_t0)
    if shared:
        L = L[:1].expand(N, -1, -1).contiguous()
    times_cpu = O.get_times(t0, cfg["dt"], T)
    tg = affine(times_cpu)
    Lg = ops.expand_shared(L[0].cuda(), N) if shared else L.cuda()
    w = ops.mvn_rsample(mean.cuda(), Lg, eps.cuda())
    traj = ops.prodmp_traj(mp, tg, w, t0.cuda(), y0.cuda(), v0.cuda())
    torch.manual_seed(1)
    pairs = O.get_time_pairs(T, dict(num_select=25, fixed_interval=True))
    call("tce_pair_env_static", int(form == "static"))
    try:
        lp = ops.pair_log_prob(mp, traj, mean.cuda(), Lg, tg, t0.cuda(),
                               y0.cuda(), v0.cuda(), pairs.cuda())
    finally:
        call("tce_pair_env_static", 1)
    n = min(N, 48)
    dd = lambda x: x[:n].double()
    truth = pair_log_prob(o64, traj.cpu()[:n].double(), dd(mean), dd(L),
                          times_cpu[:n].double(), dd(t0), dd(y0), dd(v0),
                          pairs)
    err = float((lp.cpu()[:n].double() - truth).abs().max())
    assert err <= LOGP_REL * float(truth.abs().max()), \
        (err, float(truth.abs().max()))


def test_pair_logprob_c2_size_properties(ops):
    """BASELINE C2 shape: N 4096, T 500, P 24, dof 4, nb 8 in fp32 -- oracle on
    a slice and shared-vs-per-env consistency on the whole batch."""
    name, dtype = "metaworld", torch.float32
    mp, oracle = make(name, dtype)
    N, T = 4096, 500
    mean, L, eps, t0, y0, v0 = inputs(name, N, dtype, 4)
    Lsh = L[:1]
    tg = ops.times(t0.cuda(), CFGS[name]["dt"], T)
    Lg = ops.expand_shared(Lsh[0].cuda(), N)
    w = ops.mvn_rsample(mean.cuda(), Lg, eps.cuda())
    traj = ops.prodmp_traj(mp, tg, w, t0.cuda(), y0.cuda(), v0.cuda())
    torch.manual_seed(0)
    pairs = O.get_time_pairs(T, dict(num_select=25, fixed_interval=True))
    lp_shared = ops.pair_log_prob(mp, traj, mean.cuda(), Lg, tg, t0.cuda(),
                                  y0.cuda(), v0.cuda(), pairs.cuda())
    lp_full = ops.pair_log_prob(mp, traj, mean.cuda(),
                                Lsh.expand(N, -1, -1).contiguous().cuda(), tg,
                                t0.cuda(), y0.cuda(), v0.cuda(), pairs.cuda())
    # float64 truth on slices across the batch (first / middle / ragged last
    # block): both paths within LOGP_REL of max |truth| (north_star: 1e-5)
    o64 = ProDMPOracle(dtype=torch.float64, **CFGS[name])
    worst = 0.0
    for sl in (slice(0, 16), slice(2000, 2016), slice(4080, 4096)):
        n = sl.stop - sl.start
        times_cpu = O.get_times(t0[sl], CFGS[name]["dt"], T).double()
        truth = pair_log_prob(o64, traj.cpu()[sl].double(), mean[sl].double(),
                              Lsh.expand(n, -1, -1).double(), times_cpu,
                              t0[sl].double(), y0[sl].double(),
                              v0[sl].double(), pairs)
        scale = float(truth.abs().max())
        for got in (lp_shared, lp_full):
            err = float((got.cpu()[sl].double() - truth).abs().max())
            worst = max(worst, err / scale)
            assert err <= LOGP_REL * scale, (err, scale)
    # shared-L fast path vs per-env general path: same math, different order
    scale = float(lp_full.abs().max())
    assert float((lp_shared - lp_full).abs().max()) <= 2 * LOGP_REL * scale


def test_basis_table_cache_is_invalidated_by_new_or_modified_times():
    """ops.prodmp_traj reuses the basis table only for the very same (unmodified)
    times / init-time tensors."""
    import torch
    from tce_rl_amd import ops
    from tce_rl_amd.mp import ProDMP
    mp = ProDMP(dtype=torch.float32, device="cuda", num_dof=4, num_basis=5,
                tau=5, alpha_phase=3, alpha=10, dt=0.0125,
                basis_bandwidth_factor=5, weights_scale=0.1, goal_scale=0.1,
                relative_goal=True)
    N, T = 300, 500
    g = torch.Generator(device="cuda").manual_seed(0)
    w = 0.1 * torch.randn(N, 24, device="cuda", generator=g)
    y0 = torch.rand(N, 4, device="cuda", generator=g)
    v0 = torch.zeros(N, 4, device="cuda")
    t0 = torch.zeros(N, device="cuda")
    times = ops.times(t0, mp.dt, T)
    a = ops.prodmp_traj(mp, times, w, t0, y0, v0)
    assert not (ops._times_flags(mp, torch.empty_like(times), t0) & 2)
    ops.prodmp_traj(mp, times, w, t0, y0, v0)
    assert ops._times_flags(mp, times, t0) & 2            # same objects: cached
    b = ops.prodmp_traj(mp, times, w, t0, y0, v0)
    assert torch.equal(a, b)
    # a different init time (new tensors) must rebuild the table
    t1 = torch.full((N,), 0.25, device="cuda")
    times1 = ops.times(t1, mp.dt, T)
    c = ops.prodmp_traj(mp, times1, w, t1, y0, v0)
    mp2 = ProDMP(dtype=torch.float32, device="cuda", num_dof=4, num_basis=5,
                 tau=5, alpha_phase=3, alpha=10, dt=0.0125,
                 basis_bandwidth_factor=5, weights_scale=0.1, goal_scale=0.1,
                 relative_goal=True)
    c_ref = ops.prodmp_traj(mp2, times1, w, t1, y0, v0)
    assert torch.equal(c, c_ref) and not torch.equal(a, c)
    # in-place modification bumps the version: rebuilt as well
    t1.add_(0.25)
    times2 = ops.times(t1, mp.dt, T)
    times1.copy_(times2)
    d = ops.prodmp_traj(mp, times1, w, t1, y0, v0)
    d_ref = ops.prodmp_traj(mp2, times2, w, t1, y0, v0)
    assert torch.equal(d, d_ref)


@pytest.mark.parametrize("name", ["metaworld_nb5", "box_push"])
def test_pair_kernel_forms_agree_in_float32(ops, name):
    """fp32, 1000 envs (several blocks, a ragged last one): the register form of
    the shared-covariance pair kernels against the general form."""
    from tce_rl_amd._lib import call
    dtype = torch.float32
    mp, oracle = make(name, dtype)
    N, T = 1000, HORIZON[name]
    mean, L, eps, t0, y0, v0 = inputs(name, N, dtype, 5)
    tg = ops.times(t0.cuda(), CFGS[name]["dt"], T)
    Lb = L[0].cuda()
    w = ops.mvn_rsample(mean.cuda(), ops.expand_shared(Lb, N), eps.cuda())
    traj = ops.prodmp_traj(mp, tg, w, t0.cuda(), y0.cuda(), v0.cuda())
    torch.manual_seed(0)
    pairs = O.get_time_pairs(T, dict(num_select=25, fixed_interval=True)).cuda()
    wgt = torch.randn(N, pairs.shape[0], device="cuda")
    out = []
    try:
        for form in (1, 0):
            call("tce_pair_env_static", form)
            m = mean.cuda().requires_grad_(True)
            Lg = Lb.clone().requires_grad_(True)
            lp = ops.pair_log_prob(mp, traj, m, ops.expand_shared(Lg, N), tg,
                                   t0.cuda(), y0.cuda(), v0.cuda(), pairs)
            (lp * wgt).sum().backward()
            out.append((lp.detach(), m.grad, Lg.grad))
    finally:
        call("tce_pair_env_static", 1)
    for a, b in zip(*out):
        scale = float(b.abs().max())
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-5 * scale)
