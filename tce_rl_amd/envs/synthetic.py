"""GPU-resident synthetic batched environments speaking the reference's env
protocol.

The reference steps MuJoCo worlds (fancy_gym / Metaworld) in one OS process per
env behind SB3 ``SubprocVecEnv`` (``mprl/util/util_mp.py:119-185``) -- CPU
simulators that are neither in this image nor part of the hot path.  These
stand-ins keep the *interface* the sampler consumes
(``temporal_correlated_sampler.py:226-303``): ``reset() -> obs [N, D]`` with the
tail ``[time, des_pos(dof), des_vel(dof)]`` and ``step(actions [N, T, 2 dof])
-> (next_obs, episode_reward, done, infos)`` where ``infos`` carries
``step_states [N, T, D]``, ``step_rewards``, ``step_terminations``,
``step_truncations``, ``segment_length`` and the task metrics -- as batched
device tensors instead of a list of per-env dicts.  Dynamics: the robot tracks
the desired trajectory exactly, reward = -|pos - goal|^2 - 1e-3 |vel|^2.
"""
import types

import torch

# (T, dt, task-obs dim) stand-ins per env family
_FAMILIES = {
    "metaworld": (500, 0.0125, 39),
    "BoxPushing": (100, 0.02, 20),
    "TableTennis": (350, 0.008, 19),
    "HopperJump": (250, 0.008, 15),
}


def family_of(env_id):
    for k in _FAMILIES:
        if k.lower() in env_id.lower():
            return k
    return "metaworld"


class SyntheticTCEEnv:
    def __init__(self, env_id, num_env, num_dof, dtype=torch.float32,
                 device="cuda", seed=0, num_times=None, dt=None,
                 dim_task_obs=None):
        T, dt0, d_task = _FAMILIES[family_of(env_id)]
        self.env_id, self.num_env, self.num_dof = env_id, num_env, num_dof
        self.num_times = int(num_times or T)
        self.dt = float(dt or dt0)
        self.dim_task_obs = int(dim_task_obs or max(d_task, 2 * num_dof))
        self.dtype, self.device = dtype, torch.device(device)
        self.dim_obs = self.dim_task_obs + 1 + 2 * num_dof
        self.gen = torch.Generator(device=self.device).manual_seed(seed)
        self.observation_space = types.SimpleNamespace(shape=(self.dim_obs,))
        self.action_space = types.SimpleNamespace(shape=(2 * num_dof,))
        self.spec = types.SimpleNamespace(max_episode_steps=self.num_times)
        self.envs = [self]               # sampler reads envs[0].dt / .spec
        self.event = family_of(env_id) in ("TableTennis", "HopperJump")
        self.goal = None

    def _obs(self, time, pos, vel):
        N, D = pos.shape[0], self.num_dof
        task = torch.zeros(N, self.dim_task_obs, dtype=self.dtype,
                           device=self.device)
        task[:, :D] = self.goal
        task[:, D:2 * D] = pos
        return torch.cat([task, time[:, None], pos, vel], -1)

    def reset(self):
        N, D = self.num_env, self.num_dof
        r = lambda *s: torch.rand(*s, generator=self.gen, dtype=self.dtype,
                                  device=self.device)
        self.goal = r(N, D) * 2 - 1
        pos = 0.1 * (r(N, D) * 2 - 1)
        vel = torch.zeros(N, D, dtype=self.dtype, device=self.device)
        time = torch.zeros(N, dtype=self.dtype, device=self.device)
        return self._obs(time, pos, vel)

    def step(self, actions):
        """actions [N, T, 2 dof] (desired pos | vel) -> one whole episode."""
        N, T, D = self.num_env, self.num_times, self.num_dof
        pos, vel = actions[..., :D], actions[..., D:]
        times = self.dt * torch.arange(1, T + 1, dtype=self.dtype,
                                       device=self.device)
        states = torch.zeros(N, T, self.dim_obs, dtype=self.dtype,
                             device=self.device)
        states[..., :D] = self.goal[:, None, :]
        states[..., D:2 * D] = pos
        states[..., self.dim_task_obs] = times[None, :]
        states[..., self.dim_task_obs + 1:] = actions
        rewards = -((pos - self.goal[:, None, :]) ** 2).sum(-1) \
            - 1e-3 * (vel ** 2).sum(-1)
        term = torch.zeros(N, T, dtype=torch.bool, device=self.device)
        trunc = torch.zeros(N, T, dtype=torch.bool, device=self.device)
        trunc[:, -1] = True
        dist = (pos[:, -1] - self.goal).norm(dim=-1)
        infos = {"step_states": states, "step_rewards": rewards,
                 "step_terminations": term, "step_truncations": trunc,
                 "segment_length": torch.full((N,), T, device=self.device),
                 "success": (dist < 0.05).to(self.dtype)}
        if self.event:
            # event = first step at which the hand is within 0.5 of the goal
            near = (pos - self.goal[:, None, :]).norm(dim=-1) < 0.5
            flags = torch.cummax(near.to(torch.int8), dim=1).values.bool()
            infos["hit_ball"] = flags
            infos["has_left_floor"] = flags
        next_obs = self.reset()
        done = torch.ones(N, dtype=torch.bool, device=self.device)
        return next_obs, rewards.sum(-1), done, infos


class SyntheticBBEnv(SyntheticTCEEnv):
    """Black-box (BBRL) flavour: the action is the MP parameter vector and the
    trajectory is generated *inside* the env (black_box_sampler.py:158-229)."""

    def __init__(self, env_id, num_env, mp, **kw):
        super().__init__(env_id, num_env, mp.num_dof, dtype=mp.dtype,
                         device=mp.device, **kw)
        self.mp = mp
        self.dim_obs = self.dim_task_obs
        self.observation_space = types.SimpleNamespace(shape=(self.dim_obs,))
        self.action_space = types.SimpleNamespace(shape=(mp.num_params,))
        self._pos = None

    def _obs(self, time, pos, vel):
        self._pos = pos
        return super()._obs(time, pos, vel)[:, :self.dim_task_obs]

    def step(self, params):
        from .. import ops
        N = self.num_env
        t0 = torch.zeros(N, dtype=self.dtype, device=self.device)
        v0 = torch.zeros(N, self.num_dof, dtype=self.dtype, device=self.device)
        times = ops.times(t0, self.dt, self.num_times)
        traj = ops.prodmp_traj(self.mp, times, params, t0, self._pos, v0)
        D = self.num_dof
        pos, vel = traj[..., :D], traj[..., D:]
        rewards = -((pos - self.goal[:, None, :]) ** 2).sum(-1) \
            - 1e-3 * (vel ** 2).sum(-1)
        dist = (pos[:, -1] - self.goal).norm(dim=-1)
        infos = {"trajectory_length":
                 torch.full((N,), self.num_times, device=self.device),
                 "success": (dist < 0.05).to(self.dtype)}
        next_obs = self.reset()
        done = torch.ones(N, dtype=torch.bool, device=self.device)
        return next_obs, rewards.sum(-1), done, infos


def make_env(env_id, num_env, seed, mp_args=None, black_box=False, dtype=None,
             device="cuda", **kw):
    """Stand-in for make_bb_vec_env (mprl/util/util_mp.py:155-185)."""
    if black_box:
        from ..mp import get_mp
        mp = get_mp(type="prodmp", args=dict(mp_args))
        return SyntheticBBEnv(env_id, num_env, mp, seed=seed, **kw)
    return SyntheticTCEEnv(env_id, num_env, int(mp_args["num_dof"]),
                           dtype=dtype, device=device, seed=seed, **kw)
