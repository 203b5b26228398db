"""Test doubles that speak the REFERENCE's env protocol (an SB3 vec env:
``reset() -> np [N, D]``, ``step(np actions) -> (obs, reward, done, infos)``
with ``infos`` a list of one dict of numpy values per env --
mprl/rl/sampler/temporal_correlated_sampler.py:226-303,
mprl/rl/sampler/black_box_sampler.py:200-230,
mprl/util/util_mp.py:144-185), for tests/test_vec_adapter_*.py.

* ``OracleVecEnv``: pure numpy / CPU; the episode physics are the CPU
  restatement oracle/env_oracle.py (test infrastructure), one dict per env;
* ``ReplayVecEnv``: replays a GPU ``Synthetic*Env`` through the protocol
  (device -> per-env numpy dicts), so that the adapter path can be compared BIT
  FOR BIT with the direct path on the same physics.
"""
import types

import numpy as np
import torch

from oracle import env_oracle as E


class _Space:
    def __init__(self, shape):
        self.shape = tuple(shape)


class _Spec:
    def __init__(self, T):
        self.max_episode_steps = T


def _metric_sequence(value, T):
    """A task metric as the reference's envs log it: one value per step, the
    LAST is what the sampler keeps."""
    seq = np.zeros(T, dtype=np.float64)
    seq[-1] = value
    return seq


class OracleVecEnv:
    """N point-mass envs of one family, stepped on the host (float64 numpy in,
    as MuJoCo hands out)."""

    def __init__(self, task, num_envs, dof, d_task, T, dt, seed=0):
        self.task, self.num_envs, self.dof = task, num_envs, dof
        self.d_task, self.T, self.dt = d_task, T, dt
        self.rng = np.random.default_rng(seed)
        D = d_task + 1 + 2 * dof
        self.observation_space = _Space((D,))
        self.action_space = _Space((2 * dof,))
        self._inner = types.SimpleNamespace(dt=dt, spec=_Spec(T))
        self.envs = [self._inner]
        self._obs0 = None

    def reset(self):
        N, dof = self.num_envs, self.dof
        goal = torch.from_numpy(self.rng.uniform(-1, 1, (N, dof)))
        pos = torch.from_numpy(0.1 * self.rng.uniform(-1, 1, (N, dof)))
        self._obs0 = E.reset_obs(self.task, self.d_task, goal, pos,
                                 torch.zeros_like(pos))
        return self._obs0.numpy().copy()

    def step(self, actions):
        assert isinstance(actions, np.ndarray) and \
            actions.shape == (self.num_envs, self.T, 2 * self.dof)
        a = torch.from_numpy(np.asarray(actions, dtype=np.float64))
        states, rewards, flags, metrics = E.rollout(
            self.task, a, self._obs0, self.dof, self.d_task, self.dt)
        T = self.T
        term = np.zeros(T, dtype=bool)
        trunc = np.zeros(T, dtype=bool)
        trunc[-1] = True
        infos = []
        for n in range(self.num_envs):
            d = {"step_states": states[n, 1:].numpy().copy(),
                 "step_rewards": rewards[n].numpy().copy(),
                 "step_terminations": term.copy(),
                 "step_truncations": trunc.copy(),
                 "segment_length": T,
                 "success": _metric_sequence(float(metrics[n, 0]), T),
                 "final_distance": _metric_sequence(float(metrics[n, 1]), T),
                 "not_for_the_sampler": "text"}
            if self.task in ("table_tennis", "hopper"):
                d["hit_ball"] = flags[n].numpy().copy()
                d["has_left_floor"] = flags[n].numpy().copy()
            infos.append(d)
        reward = rewards.sum(-1).numpy()
        done = np.ones(self.num_envs, dtype=bool)
        return self.reset(), reward, done, infos

    def env_method(self, name, *a, **k):
        return [None] * self.num_envs

    def get_attr(self, name):
        return [getattr(self._inner, name)] * self.num_envs

    def close(self):
        pass


class ReplayVecEnv:
    """A GPU synthetic env behind the reference protocol (numpy out, list of
    dicts); ``black_box``: the ``trajectory_length`` protocol."""

    def __init__(self, synthetic, black_box=False):
        self.syn, self.black_box = synthetic, black_box
        self.num_envs = synthetic.num_env
        self.observation_space = synthetic.observation_space
        self.action_space = synthetic.action_space
        self.envs = [types.SimpleNamespace(dt=synthetic.dt,
                                           spec=synthetic.spec)]
        self.steps_seen = 0

    def reset(self):
        return self.syn.reset().cpu().numpy()

    def step(self, actions):
        assert isinstance(actions, np.ndarray)
        self.steps_seen += 1
        a = torch.from_numpy(actions).to(self.syn.device)
        nxt, rew, done, inf = self.syn.step(a)
        host = {k: v.cpu().numpy() for k, v in inf.items()
                if torch.is_tensor(v) and k not in ("step_states_full",
                                                    "obs_moment_partials")}
        T = self.syn.num_times
        infos = []
        for n in range(self.num_envs):
            d = {}
            for k, v in host.items():
                if k in ("success", "final_distance"):
                    d[k] = _metric_sequence(v[n], T).astype(v.dtype)
                elif k in ("segment_length", "trajectory_length"):
                    d[k] = int(v[n])
                else:
                    d[k] = v[n]
            infos.append(d)
        return nxt.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy(), infos

    def env_method(self, name, *a, **k):
        return [None] * self.num_envs

    def get_attr(self, name):
        return [getattr(self.envs[0], name)] * self.num_envs

    def close(self):
        pass
