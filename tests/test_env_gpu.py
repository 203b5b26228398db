"""f1: the synthetic env suite kernel (csrc/env.hip: PD-tracked
point mass + per-family task, one launch per episode, state buffer written once
with the observation moments accumulated in the same pass) against its CPU
restatement oracle/env_oracle.py."""
import numpy as np
import pytest
import torch

from oracle import env_oracle as E
from oracle import tce_oracle as O

pytestmark = pytest.mark.gpu

# task, dof, d_task, T, dt  (the stand-in dimensions of tce_rl_amd/envs)
CASES = {
    "reach": ("reach", 4, 39, 500, 0.0125),
    "push": ("push", 7, 21, 100, 0.02),
    "push_mw": ("push", 4, 39, 500, 0.0125),
    "table_tennis": ("table_tennis", 7, 21, 350, 0.008),
    "hopper": ("hopper", 3, 17, 250, 0.008),
    "push_odd_obs": ("push", 7, 20, 100, 0.02),       # D = 35: the 4-byte store path
    # T * 2 dof = 2 mod 4: the desired trajectory ends in a half 16-byte chunk
    "push_odd_T": ("push", 7, 21, 99, 0.02),
    "hopper_odd_T": ("hopper", 3, 17, 251, 0.008),
    "push_short_odd_T": ("push", 5, 21, 17, 0.02),
}


def make_case(name, N, dtype, seed=0):
    task, dof, d_task, T, dt = CASES[name]
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g, dtype=dtype)
    goal = r(N, dof) * 2 - 1
    pos0 = 0.1 * (r(N, dof) * 2 - 1)
    vel0 = torch.zeros(N, dof, dtype=dtype)
    obs0 = E.reset_obs(task, d_task, goal, pos0, vel0)
    # desired trajectory: a smooth move from the start towards a per-env target
    # (for table tennis: the ball's path, so that some rackets do hit it) plus a
    # wiggle; velocities = its finite differences
    tt = torch.linspace(0, 1, T + 1, dtype=dtype)[None, :, None]
    target = goal.clone()
    if task == "table_tennis":
        ball0 = obs0[:, 2 * dof:2 * dof + 3]
        aim = ball0 * 0.45 * (r(N, 1) * 0.6 + 0.7)
        miss = torch.tensor([-0.8, 0.9, -0.5], dtype=dtype) * (0.5 + r(N, 1))
        target[:, :3] = torch.where((torch.arange(N) % 3 == 0)[:, None], miss,
                                    aim)       # every third racket stays away
    if task == "hopper":
        target[:, 2] = r(N) * 0.8
    s = 3 * tt ** 2 - 2 * tt ** 3
    path = pos0[:, None] + (target - pos0)[:, None] * s \
        + 0.05 * torch.sin(6.28 * tt * (1 + r(N, 1, dof)))
    des_pos = path[:, 1:]
    des_vel = (path[:, 1:] - path[:, :-1]) / dt
    return task, dof, d_task, T, dt, obs0, torch.cat([des_pos, des_vel], -1)


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_env_rollout_matches_cpu_restatement(name, dtype):
    from tce_rl_amd import ops
    N = 67
    task, dof, d_task, T, dt, obs0, actions = make_case(name, N, dtype)
    ref_s, ref_r, ref_f, ref_m = E.rollout(task, actions, obs0, dof, d_task, dt)
    shift = 0.3 * torch.randn(obs0.shape[-1], dtype=dtype)
    out = ops.env_rollout(actions.cuda(), obs0.cuda(), task, dof, d_task, dt,
                          E.KP, E.KD, want_states=True, want_flags=True,
                          shift=shift.cuda(), want_moments=True)
    tol = dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 \
        else dict(rtol=1e-11, atol=1e-11)
    assert torch.equal(out["states"][:, 0].cpu(), obs0)      # row 0 = reset obs
    torch.testing.assert_close(out["states"].cpu(), ref_s, **tol)
    torch.testing.assert_close(out["rewards"].cpu(), ref_r, **tol)
    if task in ("table_tennis", "hopper"):
        # the event is a threshold on a float: allow the envs whose margin is
        # inside rounding to flip (none in fp64)
        diff = (out["flags"].cpu() != ref_f).any(-1)
        assert diff.sum() <= (0 if dtype == torch.float64 else 1)
        assert 0 < ref_f[:, -1].sum() < N                    # both outcomes occur
    else:
        assert not out["flags"].any() and not ref_f.any()
    ok = out["metrics"][:, 0].cpu() == ref_m[:, 0]
    assert ok.sum() >= N - (0 if dtype == torch.float64 else 1)
    torch.testing.assert_close(out["metrics"][:, 1].cpu(), ref_m[:, 1], **tol)
    # moments accumulated in the same pass -> running mean / std update
    D = obs0.shape[-1]
    mean = shift.clone().cuda()
    var = torch.ones(D, dtype=dtype).cuda()
    count = ops.rms_merge(out["partials"], N * (T + 1), mean, var, 1e-4)
    rms = O.RunningMeanStd((D,), dtype)
    rms.mean = shift.clone()
    rms.update(ref_s.reshape(-1, D))
    assert count == pytest.approx(rms.count)
    mtol = dict(rtol=1e-5, atol=1e-6) if dtype == torch.float32 \
        else dict(rtol=1e-10, atol=1e-12)
    torch.testing.assert_close(mean.cpu(), rms.mean, **mtol)
    torch.testing.assert_close(var.cpu(), rms.var, **mtol)


def test_env_rollout_without_state_buffer_and_argument_checks():
    """The black-box env only needs rewards / metrics; bad shapes are refused
    with an error, not a fault."""
    from tce_rl_amd import _lib, ops
    task, dof, d_task, T, dt, obs0, actions = make_case("push_mw", 9,
                                                        torch.float32)
    ref = E.rollout(task, actions, obs0, dof, d_task, dt)
    out = ops.env_rollout(actions.cuda(), obs0.cuda(), task, dof, d_task, dt,
                          E.KP, E.KD, want_states=False)
    assert out["states"] is None and out["flags"] is None
    torch.testing.assert_close(out["rewards"].cpu(), ref[1], rtol=2e-5,
                               atol=2e-5)
    a, o = actions.cuda(), obs0.cuda()
    r = torch.empty(9, T, device="cuda")
    for bad in (dict(dof=2), dict(d_task=10), dict(family=7), dict(N=0)):
        kw = dict(family=1, N=9, dof=dof, d_task=d_task)
        kw.update(bad)
        with pytest.raises(RuntimeError, match="env_rollout"):
            _lib.call("tce_env_rollout_f32", a.data_ptr(), o.data_ptr(),
                      kw["family"], kw["N"], T, kw["dof"], kw["d_task"], dt,
                      400.0, 40.0, None, r.data_ptr(), None, None, None, None,
                      0)


def test_sampler_uses_the_fused_moments(monkeypatch):
    """TemporalCorrelatedSampler.run(training=True): observation statistics
    and normalised states equal the two-pass route (rms_update over the
    buffer, then normalise) the reference takes
    (temporal_correlated_sampler.py:244-249)."""
    from tce_rl_amd import ops
    from tce_rl_amd.config import tce_config
    from tce_rl_amd.mp_exp import MPExperiment
    cfg = tce_config("table_tennis", num_env=24, num_basis=3, epochs=1,
                     evaluation_interval=0)
    exp = MPExperiment()
    exp.initialize(cfg, 0, None)
    sampler, agent = exp.sampler, exp.agent
    raw = {}
    orig = ops.rms_normalize

    def spy(x, mean, var, eps=1e-8, inplace=False):
        raw["x"] = x.clone()
        return orig(x, mean, var, eps, inplace=inplace)
    monkeypatch.setattr(ops, "rms_normalize", spy)
    ds, n = sampler.run(training=True, policy=agent.policy, critic=agent.critic)
    assert n == 24 * 350
    x = raw["x"]                                           # [N, T+1, D] raw
    D = x.shape[-1]
    mean = torch.zeros(D, device="cuda")
    var = torch.ones(D, device="cuda")
    count = ops.rms_update(x.reshape(-1, D), mean, var, 1e-4)
    assert count == pytest.approx(sampler.obs_rms.count)
    torch.testing.assert_close(sampler.obs_rms.mean, mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(sampler.obs_rms.var, var, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(ds["step_states_full"],
                               orig(x, mean, var, 1e-8), rtol=1e-5, atol=1e-5)
    assert ds["step_states"].shape == (24, 350, D)
    assert ds["step_values"].shape == (24, 351)


class _ReferenceProtocolEnv:
    """An env that speaks only what the reference's sampler relies on
    (temporal_correlated_sampler.py:226-303): ``step(actions)`` -> 4-tuple with
    ``infos["step_states"]`` [N, T, D]; no fused moments, no whole buffer."""

    def __init__(self, inner):
        self._inner = inner

    def __getattr__(self, name):
        if name == "fused_obs_moments":
            raise AttributeError(name)
        return getattr(self._inner, name)

    def reset(self):
        return self._inner.reset()

    def step(self, actions):
        nxt, rew, done, infos = self._inner.step(actions)
        infos = {k: v for k, v in infos.items()
                 if k not in ("step_states_full", "obs_moment_partials")}
        infos["step_states"] = infos["step_states"].clone()
        return nxt, rew, done, infos


def test_sampler_accepts_an_env_without_the_fused_capability():
    """an env following the reference protocol (step(actions) ->
    step_states only) goes through cat(init_state, step_states) +
    obs_rms.update + out-of-place normalisation and yields the same dataset
    and statistics as the fused route."""
    from tce_rl_amd.config import tce_config
    from tce_rl_amd.mp_exp import MPExperiment
    out = []
    for generic in (False, True):
        cfg = tce_config("box_push", num_env=12, num_basis=3, epochs=1,
                         evaluation_interval=0)
        torch.manual_seed(5)
        exp = MPExperiment()
        exp.initialize(cfg, 0, None)
        sampler, agent = exp.sampler, exp.agent
        if generic:
            sampler.train_envs = _ReferenceProtocolEnv(sampler.train_envs)
        eps = torch.randn(12, agent.policy.dim_out,
                          generator=torch.Generator().manual_seed(1)).cuda()
        sample = agent.policy.sample
        agent.policy.sample = lambda **kw: sample(**kw, eps=eps)
        torch.manual_seed(6)
        ds, n = sampler.run(training=True, policy=agent.policy,
                            critic=agent.critic)
        out.append((ds, n, sampler.obs_rms))
    (a, na, ra), (b, nb, rb) = out
    assert na == nb == 12 * 100
    assert rb.count == pytest.approx(ra.count)
    torch.testing.assert_close(rb.mean, ra.mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rb.var, ra.var, rtol=1e-5, atol=1e-6)
    for k in ("step_states", "step_values", "step_rewards", "step_actions",
              "segment_log_prob_estimate"):
        torch.testing.assert_close(b[k], a[k], rtol=2e-5, atol=2e-5)
