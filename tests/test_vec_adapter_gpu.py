"""Row b2 (env protocol): the samplers and agents over an env that
speaks the REFERENCE's protocol -- an SB3-style vec env returning ``infos`` as a
list of one numpy dict per env (mprl/rl/sampler/temporal_correlated_sampler.py:
226-303, mprl/rl/sampler/black_box_sampler.py:200-230,
mprl/util/util_data_structure.py:310-327) -- behind
``sampler_factory(..., env_backend="vec", vec_env_fn=...)``.

* ``ReplayVecEnv`` puts the GPU synthetic env behind that protocol: the dataset
  dict of the adapter path must equal the direct path's BIT FOR BIT (reach,
  table tennis with its ``hit_ball`` event, and the black-box
  ``trajectory_length`` protocol); with the observation running mean / std on,
  the adapter path takes the reference's two-pass form (update, then
  normalise) where the direct path fuses the moments into the env kernel:
  equal to fp32 summation noise.
* ``OracleVecEnv`` is a pure-numpy env (float64 out, like MuJoCo) on the CPU
  restatement oracle/env_oracle.py: whole agent steps run over it, and its
  episodes equal the GPU env kernel's to the kernel's own tolerance.
"""
import numpy as np
import pytest
import torch

from fake_vec_env import OracleVecEnv, ReplayVecEnv

pytestmark = pytest.mark.gpu


def replay_fn(env_id, num_env, seed, render, mp_args, black_box=False, **kw):
    from tce_rl_amd.envs import make_env
    syn = make_env(env_id, num_env, seed, mp_args=mp_args, black_box=black_box,
                   dtype=torch.float32, device="cuda")
    return ReplayVecEnv(syn, black_box)


def _tce_agent(env, vec, norm, num_env=48, fn=replay_fn,
               metrics=("success", "final_distance")):
    from tce_rl_amd.config import tce_config
    from tce_rl_amd.mp_exp import MPExperiment
    cfg = tce_config(env, num_env=num_env, num_basis=5, epochs=2,
                     evaluation_interval=0, seed=3, num_env_test=8)
    sa = cfg["params"]["sampler"]["args"]
    sa["norm_step_obs"] = norm
    sa["task_specified_metrics"] = list(metrics)
    if vec:
        sa["env_backend"], sa["vec_env_fn"] = "vec", fn
    torch.manual_seed(11)
    exp = MPExperiment()
    exp.initialize(cfg, 0, None)
    return exp.agent


def _same(a, b, exact, key):
    if exact:
        assert torch.equal(a, b), key
    else:
        torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-5, msg=key)


def test_hit_ball_as_a_task_metric_keeps_the_mdp_reward_right():
    """``task_specified_metrics: [hit_ball, ...]`` as in the
    reference's table-tennis YAMLs -- the per-step flag still reaches
    make_mdp_reward in full, the logged metric is its last element, and the
    dataset equals the direct path's."""
    from tce_rl_amd import ops
    m = ("hit_ball", "success")
    direct = _tce_agent("table_tennis", False, False, metrics=m)
    vec = _tce_agent("table_tennis", True, False, metrics=m)
    out = []
    for ag in (direct, vec):
        torch.manual_seed(100)
        out.append(ag.sampler.run(training=True, policy=ag.policy,
                                  critic=ag.critic)[0])
    d0, d1 = out
    assert d0["hit_ball"].shape == d1["hit_ball"].shape == (48,)
    assert torch.equal(d0["hit_ball"], d1["hit_ball"])
    assert torch.equal(d0["step_rewards"], d1["step_rewards"])
    with pytest.raises(ValueError, match="event flags"):
        ops.mdp_reward(d0["step_rewards"], d0["hit_ball"] > 0)


@pytest.mark.parametrize("env", ["metaworld", "table_tennis"])
@pytest.mark.parametrize("norm", [False, True])
def test_tce_dataset_through_the_reference_protocol(env, norm):
    from tce_rl_amd.envs.vec_adapter import VecEnvAdapter
    direct, vec = _tce_agent(env, False, norm), _tce_agent(env, True, norm)
    assert isinstance(vec.sampler.train_envs, VecEnvAdapter)
    assert vec.sampler.dt == direct.sampler.dt
    assert vec.sampler.num_times == direct.sampler.num_times
    assert vec.sampler.observation_shape == direct.sampler.observation_shape
    for it in range(2):                     # the second rollout: auto-reset obs
        out = []
        for ag in (direct, vec):
            torch.manual_seed(100 + it)     # pair offset + policy noise
            out.append(ag.sampler.run(training=True, policy=ag.policy,
                                      critic=ag.critic))
        (d0, n0), (d1, n1) = out
        assert n0 == n1 == 48 * direct.sampler.num_times
        assert set(d0) == set(d1)
        exact = not norm
        for k in d0:
            if k == "segment_params_L":
                from tce_rl_amd import ops
                _same(ops.first_matrix(d0[k]), ops.first_matrix(d1[k]), True, k)
            elif d0[k].dtype == torch.bool or not d0[k].is_floating_point():
                assert torch.equal(d0[k], d1[k]), k
            else:
                states = k.startswith("step_states") or k == "step_values"
                _same(d0[k], d1[k], exact or not states, k)
        assert vec.sampler.train_envs.vec.steps_seen == it + 1
    if norm:                                # the same running statistics either way
        torch.testing.assert_close(vec.sampler.obs_rms.mean,
                                   direct.sampler.obs_rms.mean, rtol=1e-5,
                                   atol=1e-6)
        torch.testing.assert_close(vec.sampler.obs_rms.var,
                                   direct.sampler.obs_rms.var, rtol=1e-5,
                                   atol=1e-6)
        assert vec.sampler.obs_rms.count == pytest.approx(
            direct.sampler.obs_rms.count)


def test_tce_agent_steps_over_the_reference_protocol():
    """Whole iterations (rollout, GAE, critic + policy epochs, evaluation) with
    the env behind the list-of-dicts protocol end where the direct path ends."""
    direct, vec = _tce_agent("metaworld", False, False), \
        _tce_agent("metaworld", True, False)
    for ag in (direct, vec):
        ag.evaluation_interval = 1
        for it in range(2):
            torch.manual_seed(50 + it)
            res = ag.step()
        ag.flush_metrics()
        ag.last = dict(res.items())
    for p, q in zip(direct.policy.parameters + direct.critic.parameters,
                    vec.policy.parameters + vec.critic.parameters):
        assert torch.equal(p, q)
    for k in ("exploration_success_mean", "evaluation_success_mean",
              "critic_loss_mean", "surrogate_loss_mean", "num_global_steps"):
        assert direct.last[k] == vec.last[k], k


def test_bbrl_dataset_and_step_through_the_reference_protocol():
    """The black-box protocol: MP parameters in, ``trajectory_length`` and the
    task metrics out of the per-env dicts."""
    from tce_rl_amd.rl import sampler_factory
    from test_agent_gpu import BB_MP, build_bbrl
    agents = []
    for vec in (False, True):
        torch.manual_seed(5)
        agent, _ = build_bbrl(40, 2)
        if vec:
            mp = {"type": "prodmp", "args": dict(BB_MP, dtype="float32",
                                                 device="cuda")}
            agent.sampler = sampler_factory(
                "BlackBoxSampler", env_id="metaworld_ProDMP/push-v2",
                num_env_train=40, num_env_test=16, dtype="float32",
                device="cuda", seed=0, mp=mp,
                task_specified_metrics=["success"], env_backend="vec",
                vec_env_fn=replay_fn, env_args=dict(black_box=True))
        agents.append(agent)
    direct, vec = agents
    out = []
    for ag in agents:
        torch.manual_seed(21)
        out.append(ag.sampler.run(training=True, policy=ag.policy,
                                  critic=ag.critic))
    (d0, n0), (d1, n1) = out
    assert n0 == n1 == 40 * 500 and set(d0) == set(d1)
    for k in d0:
        if k == "segment_params_L":
            continue
        assert torch.equal(d0[k], d1[k]), k
    for ag in agents:
        torch.manual_seed(22)
        res = ag.step()
        ag.last = dict(res.items())
    for p, q in zip(direct.policy.parameters + direct.critic.parameters,
                    vec.policy.parameters + vec.critic.parameters):
        assert torch.equal(p, q)
    assert direct.last["exploration_success_mean"] == \
        vec.last["exploration_success_mean"]


def oracle_fn(env_id, num_env, seed, render, mp_args, **kw):
    task = "table_tennis" if "TableTennis" in env_id else "reach"
    T, dt, d_task = (350, 0.008, 21) if task == "table_tennis" \
        else (500, 0.0125, 39)
    return OracleVecEnv(task, num_env, int(mp_args["num_dof"]), d_task, T, dt,
                        seed=seed)


@pytest.mark.parametrize("env", ["metaworld", "table_tennis"])
def test_numpy_env_on_the_host_drives_the_gpu_agent(env):
    """A pure-numpy env (float64 list-of-dicts, stepped on the host): the agent
    trains over it, and one of its episodes equals the GPU env kernel's on the
    same reset and desired trajectory."""
    from tce_rl_amd import ops
    agent = _tce_agent(env, True, True, num_env=24, fn=oracle_fn)
    for it in range(2):
        res = agent.step()
    agent.flush_metrics()
    assert np.isfinite(res["critic_loss_mean"])
    assert np.isfinite(res["surrogate_loss_mean"])
    assert res["num_global_steps"] == 2 * 24 * agent.sampler.num_times
    assert 0.0 <= res["exploration_success_mean"] <= 1.0
    # one more rollout by hand: the numpy env's episode vs the env kernel
    sam = agent.sampler
    vec = sam.train_envs.vec
    obs0 = torch.from_numpy(vec.reset())
    g = torch.Generator().manual_seed(1)
    dof = vec.dof
    tt = torch.linspace(0, 1, vec.T + 1, dtype=torch.float64)[None, :, None]
    path = obs0[:, None, :dof] + 0.3 * tt * torch.rand(24, 1, dof, generator=g,
                                                       dtype=torch.float64)
    acts = torch.cat([path[:, 1:], (path[:, 1:] - path[:, :-1]) / vec.dt], -1)
    _, rew, _, infos = vec.step(acts.numpy())
    out = ops.env_rollout(acts.float().cuda(), obs0.float().cuda(), vec.task,
                          dof, vec.d_task, vec.dt, 400.0, 40.0,
                          want_states=True, want_flags=True)
    host_states = np.stack([d["step_states"] for d in infos])
    np.testing.assert_allclose(out["states"][:, 1:].cpu().numpy(), host_states,
                               rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(out["rewards"].sum(-1).cpu().numpy(), rew,
                               rtol=2e-4, atol=2e-4)
