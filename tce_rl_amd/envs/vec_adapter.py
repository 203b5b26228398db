"""The reference's env protocol -> what the GPU samplers index (SURVEY 8b "Env
protocol").

The reference steps an SB3 vec env (``make_bb_vec_env``,
mprl/util/util_mp.py:144-185: ``SubprocVecEnv`` / ``DummyVecEnv`` over
fancy_gym black-box envs): ``reset() -> np [N, D]`` and ``step(np actions)
-> (next_obs [N, D], reward [N], done [N], infos)`` with ``infos`` a LIST of
one dict per env holding numpy values --

* TCE (mprl/rl/sampler/temporal_correlated_sampler.py:226-303):
  ``step_states [T, D]`` (without the initial state), ``step_rewards [T]``,
  ``step_terminations [T]``, ``step_truncations [T]``, ``segment_length``, the
  task metrics (a per-step sequence whose LAST element is logged:
  ``get_item_from_dicts(infos, metric, lambda x: x[-1])``,
  mprl/util/util_data_structure.py:310-327) and the event flags ``hit_ball`` /
  ``has_left_floor [T]`` (mprl/util/util_experiment.py:290-300);
* black box (mprl/rl/sampler/black_box_sampler.py:200-230):
  ``trajectory_length`` and the task metrics.

``VecEnvAdapter`` wraps ANY object with that surface and hands the samplers
what they index: one dict of batched device tensors.  Per key the per-env
values are stacked straight into ONE pinned staging buffer (no intermediate
``np.asarray`` of a list of arrays) and go up with ONE host -> device copy per
key and episode; the actions come down with one copy.  ``step_states`` stays
[N, T, D] without the initial row, so the sampler takes the branch that
prepends it exactly as the reference does (:233-240).  MuJoCo / fancy_gym are
not in this image and are not rebuilt: ``make_bb_vec_env`` below builds the
reference's vec env when those packages are importable, any callable with its
signature can be passed instead (``vec_env_fn``).
"""
import importlib
import os
import types

import numpy as np
import torch


_POOL_THREADS = max(1, min(8, (os.cpu_count() or 1)))
_POOL = []


def _pool():
    if not _POOL:
        from concurrent.futures import ThreadPoolExecutor
        _POOL.append(ThreadPoolExecutor(max_workers=_POOL_THREADS))
    return _POOL[0]


class VecEnvAdapter:
    # the sampler's reference-protocol branch (no fused observation moments)
    fused_obs_moments = False

    # per-step event flags make_mdp_reward reads in full
    # (mprl/util/util_experiment.py:290-300): never reduced here, even when a
    # config also lists them as task metrics
    # (mprl/config/table_tennis_4d/tcp/entire/shared.yaml:136) -- the sampler
    # takes the last element for the log itself
    EVENT_KEYS = ("hit_ball", "has_left_floor")

    def __init__(self, vec_env, dtype=torch.float32, device="cuda",
                 last_element_keys=()):
        """last_element_keys: info keys whose per-env value is a per-step
        sequence of which only the LAST element is wanted (the task metrics);
        EVENT_KEYS always come up as the full per-step array."""
        self.vec = vec_env
        self.num_env = int(vec_env.num_envs)
        self.dtype, self.device = dtype, torch.device(device)
        self.last_element_keys = tuple(last_element_keys or ())
        self.observation_space = vec_env.observation_space
        self.action_space = vec_env.action_space
        self._pinned = {}                 # key -> [host tensor, event of its last upload]
        self._np_dtype = {torch.float32: np.float32,
                          torch.float64: np.float64}[dtype]

    # ---- what the samplers read off the debug env ----------------------------
    @property
    def envs(self):
        """``debug_env.envs[0].dt`` / ``.spec.max_episode_steps``
        (temporal_correlated_sampler.py:40-41): DummyVecEnv has ``envs``, a
        SubprocVecEnv answers ``get_attr``."""
        inner = getattr(self.vec, "envs", None)
        if inner:
            return inner
        dt = self.vec.get_attr("dt")[0]
        spec = self.vec.get_attr("spec")[0]
        return [types.SimpleNamespace(dt=dt, spec=spec)]

    @property
    def spec(self):
        return self.envs[0].spec

    def env_method(self, *args, **kwargs):
        return self.vec.env_method(*args, **kwargs)

    def get_attr(self, *args, **kwargs):
        return self.vec.get_attr(*args, **kwargs)

    def render(self, *args, **kwargs):
        return self.vec.render(*args, **kwargs)

    def close(self):
        return self.vec.close()

    # ---- host <-> device -------------------------------------------------------
    def _stage(self, key, shape, np_dtype):
        """The pinned staging buffer of `key` as a numpy view (its previous
        upload has finished before it is rewritten)."""
        slot = self._pinned.get(key)
        tdtype = torch.from_numpy(np.empty(0, dtype=np_dtype)).dtype
        if slot is None or tuple(slot[0].shape) != tuple(shape) or \
                slot[0].dtype != tdtype:
            host = torch.empty(tuple(shape), dtype=tdtype)
            if self.device.type == "cuda":
                host = host.pin_memory()
            slot = self._pinned[key] = [host, None]
        if slot[1] is not None:
            slot[1].synchronize()
            slot[1] = None
        return slot, slot[0].numpy()

    def _upload(self, slot):
        if self.device.type != "cuda":
            return slot[0].clone()
        dev = slot[0].to(self.device, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record()
        return dev

    def _stack(self, key, values, kind=None):
        """N per-env values -> one device tensor [N, ...]: stacked directly into
        the pinned buffer (cast to the sampler's dtype on the way), one copy."""
        first = np.asarray(values[0])
        if kind is None:
            kind = np.bool_ if first.dtype == np.bool_ else (
                np.int64 if np.issubdtype(first.dtype, np.integer)
                else self._np_dtype)
        slot, view = self._stage(key, (len(values),) + first.shape, kind)
        if first.ndim == 0:
            view[...] = np.asarray(values, dtype=kind)
        elif view.nbytes >= (8 << 20) and len(values) >= 64:
            # the big keys (step_states: 400 MB at the headline size) in slices on
            # a few threads: the copy + cast loops release the GIL (4096 envs x
            # [500, 48] float64 -> float32: 40 ms on one thread)
            n, k = len(values), _POOL_THREADS
            cuts = [n * i // k for i in range(k + 1)]
            list(_pool().map(
                lambda ab: np.stack(values[ab[0]:ab[1]], axis=0,
                                    out=view[ab[0]:ab[1]], casting="unsafe"),
                [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]))
        else:
            np.stack(values, axis=0, out=view, casting="unsafe")
        return self._upload(slot)

    def _array(self, key, arr, kind=None):
        arr = np.asarray(arr)
        if kind is None:
            kind = np.bool_ if arr.dtype == np.bool_ else self._np_dtype
        slot, view = self._stage(key, arr.shape, kind)
        np.copyto(view, arr, casting="unsafe")
        return self._upload(slot)

    # ---- the protocol ------------------------------------------------------------
    def reset(self):
        return self._array("__obs", self.vec.reset())

    def step(self, actions):
        """actions: device tensor [N, T, 2 dof] (TCE) or [N, K] (black box)."""
        if self.device.type == "cuda":
            # (down through a pinned buffer: a pageable device -> host copy goes
            # through the runtime's bounce buffers at a fraction of the link rate)
            slot = self._pinned.get("__actions")
            if slot is None or tuple(slot[0].shape) != tuple(actions.shape) \
                    or slot[0].dtype != actions.dtype:
                slot = self._pinned["__actions"] = [
                    torch.empty(tuple(actions.shape),
                                dtype=actions.dtype).pin_memory(), None]
            slot[0].copy_(actions.detach(), non_blocking=True)
            torch.cuda.current_stream().synchronize()
            a = slot[0].numpy()
        else:
            a = actions.detach().to("cpu").numpy()
        obs, reward, done, infos = self.vec.step(a)
        if len(infos) != self.num_env:
            raise RuntimeError("VecEnvAdapter: %d info dicts for %d envs"
                               % (len(infos), self.num_env))
        out = {}
        keys = [k for k in infos[0] if all(k in d for d in infos)]
        for k in keys:
            v0 = infos[0][k]
            if k in self.last_element_keys and k not in self.EVENT_KEYS:
                out[k] = self._stack(k, [np.asarray(d[k])[-1] for d in infos])
            elif isinstance(v0, (np.ndarray, list, tuple, float, int, bool,
                                 np.generic)):
                try:
                    out[k] = self._stack(k, [d[k] for d in infos])
                except (ValueError, TypeError):
                    continue              # ragged / non-numeric: not for the sampler
        # the env-step count without a device -> host read (the reference sums
        # it on the host as well: temporal_correlated_sampler.py:296-298)
        for k in ("segment_length", "trajectory_length"):
            if k in keys:
                out["num_steps_host"] = int(
                    np.asarray([d[k] for d in infos]).sum())
        return (self._array("__obs", obs), self._array("__reward", reward),
                self._array("__done", done, np.bool_), out)


def resolve_callable(spec):
    """A callable, or "package.module:function" (so a YAML config can name it)."""
    if callable(spec):
        return spec
    mod, _, name = str(spec).partition(":")
    return getattr(importlib.import_module(mod), name)


def make_bb_vec_env(env_id, num_env, seed, render, mp_args, **kwargs):
    """The reference's vec env itself (mprl/util/util_mp.py:119-185) where its
    packages exist: fancy_gym black-box envs behind SB3's SubprocVecEnv (one
    process per env) / DummyVecEnv.  Not importable in the build image: the
    ImportError says so instead of substituting anything."""
    try:
        import gymnasium as gym
        import fancy_gym                                   # noqa: F401
        from stable_baselines3.common.vec_env import DummyVecEnv, SubprocVecEnv
    except ImportError as e:
        raise ImportError(
            "env_backend='vec' with the default vec_env_fn needs gymnasium, "
            "fancy_gym and stable_baselines3 (the reference's env stack, "
            "conda_env.sh); pass vec_env_fn=<callable or 'module:function'> "
            "returning an SB3-style vec env otherwise") from e
    if render:
        assert num_env == 1, "Rendering only works with num_env=1"
    override = _override_mp_config(mp_args)

    def make(rank):
        def _get():
            env = gym.make(id=env_id, render_mode="human" if render else None,
                           mp_config_override=override, **kwargs)
            env.reset(seed=seed + rank)
            return env
        return _get
    cls = SubprocVecEnv if num_env > 1 else DummyVecEnv
    return cls([make(i) for i in range(num_env)])


def _override_mp_config(mp_args):
    """get_override_mp_config (mprl/util/util_mp.py:60-116): the MP
    hyper-parameters of the experiment config as fancy_gym's override dict."""
    cfg = {"phase_generator_kwargs": {}, "basis_generator_kwargs": {},
           "trajectory_generator_kwargs": {}}
    ph, bs, tg = cfg["phase_generator_kwargs"], cfg["basis_generator_kwargs"], \
        cfg["trajectory_generator_kwargs"]
    for k in ("tau", "delay", "learn_tau", "learn_delay", "alpha_phase",
              "alpha"):
        if k in mp_args:
            ph[k] = mp_args[k]
    for k in ("num_basis", "basis_bandwidth_factor", "num_basis_outside"):
        if k in mp_args:
            bs[k] = mp_args[k]
    for k in ("disable_goal", "relative_goal", "auto_scale_basis",
              "weights_scale", "goal_scale"):
        if k in mp_args:
            tg[k] = mp_args[k]
    if "verbose_level" in mp_args:
        cfg.setdefault("black_box_kwargs", {})["verbose_level"] = \
            mp_args["verbose_level"]
    return cfg
