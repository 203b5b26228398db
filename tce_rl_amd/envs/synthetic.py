"""GPU-resident synthetic batched environments speaking the reference's env
protocol.

The reference steps MuJoCo worlds (fancy_gym / Metaworld) in one OS process per
env behind SB3 ``SubprocVecEnv`` (``mprl/util/util_mp.py:119-185``) -- CPU
simulators that are neither in this image nor part of the hot path.  These
stand-ins keep the *interface* the sampler consumes
(``temporal_correlated_sampler.py:226-303``): ``reset() -> obs [N, D]`` with the
tail ``[time, des_pos(dof), des_vel(dof)]`` and ``step(actions [N, T, 2 dof])
-> (next_obs, episode_reward, done, infos)`` where ``infos`` carries
``step_states``, ``step_rewards``, ``step_terminations``, ``step_truncations``,
``segment_length``, the task metrics and the ``hit_ball`` / ``has_left_floor``
event flags -- as batched device tensors instead of a list of per-env dicts.

A whole episode is ONE HIP kernel (``csrc/env.hip``): a unit point mass per
degree of freedom tracks the desired trajectory under PD control, and a small
task per env family sits on top of it:

* ``reach`` (Metaworld-like): hand = q[:3] to a goal; r = -|hand - goal|^2;
* ``push`` (Metaworld push / box pushing): an object the hand carries along
  once it is within 0.1 of it; dense r = -|obj - goal|^2 - 0.1 |hand - obj|^2;
* ``table_tennis``: a ball flying towards the origin; the step at which the
  racket comes within 0.2 of it is the ``hit_ball`` event, afterwards the ball
  leaves with the racket's velocity and r = -|ball_xy - goal_xy|^2;
* ``hopper``: reach-like, the ``has_left_floor`` event is q[2] > 0.3;
all with the velocity penalty -1e-3 |qd|^2.  The kernel writes the
``[N, T+1, D]`` state buffer once (initial observation in row 0) and
accumulates the observation running-mean/std moments in the same pass.
"""
import types

import torch

# (T, dt, task-obs dim, family) stand-ins per env family
_FAMILIES = {
    "metaworld": (500, 0.0125, 39, "reach"),
    "BoxPushing": (100, 0.02, 21, "push"),             # task dims chosen so that the
    "TableTennis": (350, 0.008, 21, "table_tennis"),   # obs dim D is a multiple of 4
    "HopperJump": (250, 0.008, 17, "hopper"),          # (16-byte row stores)
}
KP, KD = 400.0, 40.0            # critically damped tracking, stable for dt <= 0.02


def family_of(env_id):
    for k in _FAMILIES:
        if k.lower() in env_id.lower():
            return k
    return "metaworld"


def task_of(env_id):
    fam = _FAMILIES[family_of(env_id)][3]
    if fam == "reach" and "push" in env_id.lower():
        return "push"                       # metaworld push-v2 (BBRL config)
    return fam


_TT_BASE = {}


def initial_object(task, goal3, hand):
    """Object / ball position at reset as a function of goal and hand (no
    extra random draw, so a forced (goal, pos) reset fixes the whole state)."""
    if task == "table_tennis":
        key = (hand.dtype, hand.device)
        base = _TT_BASE.get(key)
        if base is None:                 # (uploaded once: a host -> device copy
            base = _TT_BASE[key] = torch.tensor(   # per reset would wait for the stream)
                [1.5, 0.0, 0.3], dtype=hand.dtype, device=hand.device)
        return base + 0.2 * goal3
    return hand + 0.25 * (goal3 - hand)


class SyntheticTCEEnv:
    # step(actions, obs_shift=..., want_moments=True) returns the whole
    # [N, T+1, D] buffer and the column moments of the same pass; the sampler
    # takes the reference protocol's path for envs without this attribute
    fused_obs_moments = True

    def __init__(self, env_id, num_env, num_dof, dtype=torch.float32,
                 device="cuda", seed=0, num_times=None, dt=None,
                 dim_task_obs=None):
        T, dt0, d_task, _ = _FAMILIES[family_of(env_id)]
        self.env_id, self.num_env, self.num_dof = env_id, num_env, num_dof
        self.task = task_of(env_id)
        self.num_times = int(num_times or T)
        self.dt = float(dt or dt0)
        self.dim_task_obs = int(dim_task_obs or max(d_task, 2 * num_dof + 6))
        self.dtype, self.device = dtype, torch.device(device)
        self.dim_obs = self.dim_task_obs + 1 + 2 * num_dof
        if not 3 <= num_dof <= 8 or self.dim_obs > 64:
            raise NotImplementedError("synthetic env suite: 3 <= num_dof <= 8 "
                                      "and obs dim <= 64")
        self.gen = torch.Generator(device=self.device).manual_seed(seed)
        self.observation_space = types.SimpleNamespace(shape=(self.dim_obs,))
        self.action_space = types.SimpleNamespace(shape=(2 * num_dof,))
        self.spec = types.SimpleNamespace(max_episode_steps=self.num_times)
        self.envs = [self]               # sampler reads envs[0].dt / .spec
        self.event = self.task in ("table_tennis", "hopper")
        self.goal = None                 # [N, dof]; the task uses goal[:, :3]
        self._obs0 = None                # observation of the current reset

    def _obs(self, time, pos, vel):
        """[q | qd | obj | goal | 0.. | time | des_pos | des_vel]."""
        N, D = pos.shape[0], self.num_dof
        task = torch.zeros(N, self.dim_task_obs, dtype=self.dtype,
                           device=self.device)
        g3 = self.goal[:, :3]
        task[:, :D] = pos
        task[:, D:2 * D] = vel
        task[:, 2 * D:2 * D + 3] = initial_object(self.task, g3, pos[:, :3])
        task[:, 2 * D + 3:2 * D + 6] = g3
        self._obs0 = torch.cat([task, time[:, None], pos, vel], -1)
        return self._obs0

    def reset(self):
        """goal ~ U(-1, 1)^dof, hand position ~ 0.1 U(-1, 1)^dof, at rest, time
        0 -- as few launches as it takes (a rollout resets twice and its ~50
        small launches run at the device's back-to-back launch interval): one
        draw for both, the observation written into one zeroed buffer (_obs
        builds the same thing for a forced (goal, pos, vel) reset)."""
        N, D = self.num_env, self.num_dof
        u = torch.empty(N, 2 * D, dtype=self.dtype, device=self.device) \
            .uniform_(-1.0, 1.0, generator=self.gen)
        self.goal = u[:, :D]
        obs = torch.zeros(N, self.dim_obs, dtype=self.dtype, device=self.device)
        pos = obs[:, :D]
        torch.mul(u[:, D:], 0.1, out=pos)
        g3 = self.goal[:, :3]
        obs[:, 2 * D:2 * D + 3] = initial_object(self.task, g3, pos[:, :3])
        obs[:, 2 * D + 3:2 * D + 6] = g3
        obs[:, self.dim_task_obs + 1:self.dim_task_obs + 1 + D] = pos
        self._obs0 = obs
        return self._visible(obs)

    def _visible(self, obs):
        """What reset() hands out of the full observation."""
        return obs

    def step(self, actions, obs_shift=None, want_moments=False):
        """actions [N, T, 2 dof] (desired pos | vel) -> one whole episode.
        obs_shift / want_moments: also return the column moments of the
        state buffer (``infos["obs_moment_partials"]``, relative to obs_shift)
        for the sampler's running mean/std."""
        from .. import ops
        N, T = self.num_env, self.num_times
        out = ops.env_rollout(actions, self._obs0, self.task, self.num_dof,
                              self.dim_task_obs, self.dt, KP, KD,
                              want_states=True, want_flags=self.event,
                              shift=obs_shift, want_moments=want_moments)
        full = out["states"]                        # [N, T+1, D], row 0 = reset obs
        term = torch.zeros(N, T, dtype=torch.bool, device=self.device)
        trunc = torch.zeros(N, T, dtype=torch.bool, device=self.device)
        trunc[:, -1] = True
        infos = {"step_states": full[:, 1:], "step_states_full": full,
                 "step_rewards": out["rewards"],
                 "step_terminations": term, "step_truncations": trunc,
                 "segment_length": torch.full((N,), T, device=self.device),
                 "num_steps_host": N * T,
                 "success": out["metrics"][:, 0],
                 "final_distance": out["metrics"][:, 1],
                 "obs_moment_partials": out["partials"]}
        if self.event:
            infos["hit_ball"] = out["flags"]
            infos["has_left_floor"] = out["flags"]
        episode_reward = out["rewards"].sum(-1)
        next_obs = self.reset()
        done = torch.ones(N, dtype=torch.bool, device=self.device)
        return next_obs, episode_reward, done, infos


class SyntheticBBEnv(SyntheticTCEEnv):
    """Black-box (BBRL) flavour: the action is the MP parameter vector and the
    trajectory is generated *inside* the env (black_box_sampler.py:158-229);
    the observation is the task part only."""

    def __init__(self, env_id, num_env, mp, **kw):
        super().__init__(env_id, num_env, mp.num_dof, dtype=mp.dtype,
                         device=mp.device, **kw)
        self.mp = mp
        self.observation_space = types.SimpleNamespace(
            shape=(self.dim_task_obs,))
        self.action_space = types.SimpleNamespace(shape=(mp.num_params,))

    def _obs(self, time, pos, vel):
        return super()._obs(time, pos, vel)[:, :self.dim_task_obs]

    def _visible(self, obs):
        return obs[:, :self.dim_task_obs]

    def step(self, params):
        from .. import ops
        N, D = self.num_env, self.num_dof
        full0 = self._obs0
        t0 = torch.zeros(N, dtype=self.dtype, device=self.device)
        times = ops.times(t0, self.dt, self.num_times)
        traj = ops.prodmp_traj(self.mp, times, params, t0, full0[:, :D],
                               full0[:, D:2 * D])
        out = ops.env_rollout(traj, full0, self.task, D, self.dim_task_obs,
                              self.dt, KP, KD, want_states=False)
        infos = {"trajectory_length":
                 torch.full((N,), self.num_times, device=self.device),
                 "num_steps_host": N * self.num_times,
                 "success": out["metrics"][:, 0],
                 "final_distance": out["metrics"][:, 1]}
        episode_reward = out["rewards"].sum(-1)
        next_obs = self.reset()
        done = torch.ones(N, dtype=torch.bool, device=self.device)
        return next_obs, episode_reward, done, infos


def make_env(env_id, num_env, seed, mp_args=None, black_box=False, dtype=None,
             device="cuda", **kw):
    """Stand-in for make_bb_vec_env (mprl/util/util_mp.py:155-185)."""
    if black_box:
        from ..mp import get_mp
        mp = get_mp(type="prodmp", args=dict(mp_args))
        return SyntheticBBEnv(env_id, num_env, mp, seed=seed, **kw)
    return SyntheticTCEEnv(env_id, num_env, int(mp_args["num_dof"]),
                           dtype=dtype, device=device, seed=seed, **kw)
