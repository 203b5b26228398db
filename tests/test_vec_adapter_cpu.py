"""Row b2 (env protocol), host side: envs/vec_adapter.VecEnvAdapter turns the
reference's ``infos: list[dict]`` of per-env numpy values
(mprl/rl/sampler/temporal_correlated_sampler.py:226-303,
mprl/util/util_data_structure.py:310-327) into the dict of batched tensors the
samplers index.  CPU tensors here; tests/test_vec_adapter_gpu.py runs the
samplers and agents over it."""
import numpy as np
import pytest
import torch

from fake_vec_env import OracleVecEnv
from tce_rl_amd.envs.vec_adapter import (VecEnvAdapter, make_bb_vec_env,
                                         resolve_callable,
                                         _override_mp_config)


def _adapter(task="table_tennis", N=5, dtype=torch.float32,
             keys=("success", "final_distance")):
    vec = OracleVecEnv(task, N, dof=7, d_task=21, T=30, dt=0.008, seed=3)
    return vec, VecEnvAdapter(vec, dtype=dtype, device="cpu",
                              last_element_keys=list(keys))


def test_event_flags_listed_as_metrics_stay_per_step():
    """The reference's table-tennis configs list ``hit_ball`` among the task
    metrics (mprl/config/table_tennis_4d/tcp/entire/shared.yaml:136) while
    make_mdp_reward reads the whole per-step flag
    (mprl/util/util_experiment.py:290-300): the adapter must not reduce it."""
    vec, ad = _adapter(keys=("hit_ball", "success"))
    ad.reset()
    g = torch.Generator().manual_seed(0)
    actions = 0.1 * torch.randn(5, 30, 14, generator=g)
    infos = ad.step(actions)[3]
    assert infos["hit_ball"].shape == (5, 30)
    assert infos["hit_ball"].dtype == torch.bool
    assert infos["success"].shape == (5,)
    from tce_rl_amd.rl.sampler import _last_element
    assert torch.equal(_last_element(infos["hit_ball"]),
                       infos["hit_ball"][:, -1])
    assert _last_element(infos["success"]) is infos["success"]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_list_of_dicts_becomes_batched_tensors(dtype):
    vec, ad = _adapter(dtype=dtype)
    N, T, D = 5, 30, 21 + 1 + 14
    obs = ad.reset()
    assert obs.shape == (N, D) and obs.dtype == dtype
    raw0 = vec._obs0.clone()
    g = torch.Generator().manual_seed(0)
    actions = 0.1 * torch.randn(N, T, 14, generator=g, dtype=dtype)
    nxt, rew, done, infos = ad.step(actions)
    # what the reference would have stacked itself (get_item_from_dicts + to_ts)
    from oracle import env_oracle as E
    s, r, f, m = E.rollout("table_tennis", actions.double(), raw0, 7, 21, 0.008)
    assert infos["step_states"].shape == (N, T, D)          # no initial row
    assert infos["step_states"].dtype == dtype
    torch.testing.assert_close(infos["step_states"], s[:, 1:].to(dtype))
    torch.testing.assert_close(infos["step_rewards"], r.to(dtype))
    assert infos["hit_ball"].dtype == torch.bool
    assert torch.equal(infos["hit_ball"], f)
    assert infos["step_terminations"].dtype == torch.bool
    assert not infos["step_terminations"].any()
    assert infos["step_truncations"][:, -1].all()
    assert infos["segment_length"].dtype == torch.int64
    assert infos["segment_length"].tolist() == [T] * N
    assert infos["num_steps_host"] == N * T                 # a host int
    # task metrics: the LAST element of the per-step sequence
    assert infos["success"].shape == (N,)
    torch.testing.assert_close(infos["success"], m[:, 0].to(dtype))
    torch.testing.assert_close(infos["final_distance"], m[:, 1].to(dtype))
    assert "not_for_the_sampler" not in infos
    assert nxt.shape == (N, D) and rew.shape == (N,) and done.dtype == torch.bool
    torch.testing.assert_close(rew, r.sum(-1).to(dtype))
    # the staging buffers are reused, the tensors handed out are not
    keep = infos["step_rewards"].clone()
    ad.step(actions * 0.5)
    assert torch.equal(infos["step_rewards"], keep)


def test_debug_env_surface_and_errors():
    vec, ad = _adapter()
    assert ad.envs[0].dt == 0.008
    assert ad.envs[0].spec.max_episode_steps == 30 and ad.spec is ad.envs[0].spec
    assert ad.observation_space.shape == (36,)
    vec.envs = []                          # a SubprocVecEnv has no .envs: get_attr
    assert ad.envs[0].dt == 0.008 and ad.envs[0].spec.max_episode_steps == 30
    ad.reset()
    vec.num_envs = 4                       # one dict short of what was promised
    bad = VecEnvAdapter(vec, device="cpu")
    vec.num_envs = 5
    with pytest.raises(RuntimeError, match="info dicts"):
        bad.step(torch.zeros(5, 30, 14))


def test_callable_by_name_and_missing_env_stack():
    assert resolve_callable("fake_vec_env:OracleVecEnv") is OracleVecEnv
    assert resolve_callable(len) is len
    # the reference's own env stack is not in this image: say so, substitute nothing
    with pytest.raises(ImportError, match="fancy_gym"):
        make_bb_vec_env("metaworld_ProDMP_TCE/reach-v2", 2, 0, False, {})
    cfg = _override_mp_config(dict(tau=5.0, num_basis=5, alpha=10,
                                   relative_goal=True, weights_scale=0.1))
    assert cfg["phase_generator_kwargs"] == {"tau": 5.0, "alpha": 10}
    assert cfg["basis_generator_kwargs"] == {"num_basis": 5}
    assert cfg["trajectory_generator_kwargs"] == {"relative_goal": True,
                                                  "weights_scale": 0.1}


def test_sampler_rejects_unknown_backend():
    from tce_rl_amd.rl.sampler import BlackBoxSampler
    with pytest.raises(ValueError, match="env_backend"):
        BlackBoxSampler("x", env_backend="gym", device="cpu")
