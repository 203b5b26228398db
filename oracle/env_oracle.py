"""CPU restatement of the synthetic env suite (tce_rl_amd/envs/synthetic.py +
csrc/env.hip): PD-tracked point mass + per-family task, one Python loop over
the T steps, vectorised over envs.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  The reference's envs
(fancy_gym / Metaworld / MuJoCo, mprl/util/util_mp.py:119-185) are third-party
and absent; this suite is the build's own stand-in that speaks the same
per-episode protocol (mprl/rl/sampler/temporal_correlated_sampler.py:226-303),
so there is no reference arithmetic to pin here -- the check is kernel == this
restatement.
"""
import torch

KP, KD = 400.0, 40.0
FAMILY = {500: ("reach", 39), 100: ("push", 21), 350: ("table_tennis", 21),
          250: ("hopper", 17)}


def initial_object(task, goal3, hand):
    if task == "table_tennis":
        base = torch.tensor([1.5, 0.0, 0.3], dtype=hand.dtype)
        return base + 0.2 * goal3
    return hand + 0.25 * (goal3 - hand)


def reset_obs(task, d_task, goal, pos, vel):
    """[q | qd | obj | goal3 | 0.. | time = 0 | des_pos | des_vel]."""
    N, dof = pos.shape
    t = torch.zeros(N, d_task, dtype=pos.dtype)
    g3 = goal[:, :3]
    t[:, :dof] = pos
    t[:, dof:2 * dof] = vel
    t[:, 2 * dof:2 * dof + 3] = initial_object(task, g3, pos[:, :3])
    t[:, 2 * dof + 3:2 * dof + 6] = g3
    return torch.cat([t, torch.zeros(N, 1, dtype=pos.dtype), pos, vel], -1)


def rollout(task, actions, obs0, dof, d_task, dt, kp=KP, kd=KD):
    """-> states [N, T+1, D], rewards [N, T], flags [N, T] bool,
    metrics [N, 2] {success, final distance}."""
    N, T, _ = actions.shape
    dtype = actions.dtype
    D = d_task + 1 + 2 * dof
    q, qd = obs0[:, :dof].clone(), obs0[:, dof:2 * dof].clone()
    obj = obs0[:, 2 * dof:2 * dof + 3].clone()
    goal = obs0[:, 2 * dof + 3:2 * dof + 6]
    ov = -obj / (float(T) * dt) if task == "table_tennis" \
        else torch.zeros_like(obj)
    hp = q[:, :3].clone()
    event = torch.zeros(N, dtype=torch.bool)
    states = torch.zeros(N, T + 1, D, dtype=dtype)
    states[:, 0] = obs0
    rewards = torch.zeros(N, T, dtype=dtype)
    flags = torch.zeros(N, T, dtype=torch.bool)
    dist2 = torch.zeros(N, dtype=dtype)
    sq = lambda x: (x * x).sum(-1)
    for i in range(T):
        dp, dv = actions[:, i, :dof], actions[:, i, dof:]
        a = kp * (dp - q) + kd * (dv - qd)
        qd = qd + dt * a
        q = q + dt * qd
        h = q[:, :3]
        v2 = sq(qd)
        if task == "push":
            touch = sq(hp - obj) < 0.01
            obj = torch.where(touch[:, None], obj + (h - hp), obj)
            g2 = sq(obj - goal)
            dist2 = g2
            rew = -g2 - 0.1 * sq(h - obj) - 1e-3 * v2
        elif task == "table_tennis":
            obj = obj + dt * ov
            b2 = sq(h - obj)
            hit_now = (~event) & (b2 < 0.04)
            event = event | hit_now
            ov = torch.where(hit_now[:, None], qd[:, :3], ov)
            g2 = sq(obj[:, :2] - goal[:, :2])
            dist2 = g2
            rew = torch.where(event, -g2, -b2) - 1e-3 * v2
        else:
            g2 = sq(h - goal)
            dist2 = g2
            rew = -g2 - 1e-3 * v2
            if task == "hopper":
                event = event | (h[:, 2] > 0.3)
        hp = h.clone()
        row = states[:, i + 1]
        row[:, :dof], row[:, dof:2 * dof] = q, qd
        row[:, 2 * dof:2 * dof + 3] = obj
        row[:, 2 * dof + 3:2 * dof + 6] = goal
        row[:, d_task] = float(i + 1) * dt
        row[:, d_task + 1:] = actions[:, i]
        rewards[:, i] = rew
        flags[:, i] = event
    lim = 0.09 if task == "table_tennis" else 0.0025
    ok = dist2 < lim
    if task == "table_tennis":
        ok = ok & event
    metrics = torch.stack([ok.to(dtype), dist2.sqrt()], -1)
    return states, rewards, flags, metrics
