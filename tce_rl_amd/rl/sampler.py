"""Rollout samplers: mirror of mprl/rl/sampler/ (abstract_sampler.py:6-43,
black_box_sampler.py:14-249, temporal_correlated_sampler.py:17-356).

Same constructor kwargs and ``run`` contract: ``run(training, policy, critic,
deterministic=False, render=False, task_specified_metrics=None) ->
(dataset dict, num_env_steps)``.  The whole rollout buffer stays on the device:
no host copies between the policy, the (synthetic) env, the critic and the
dataset (the reference crosses host <-> device and a process boundary at
temporal_correlated_sampler.py:229-240).
"""
from abc import ABC, abstractmethod

import torch

from .. import ops, util
from ..envs import make_env
from ..util import assert_shape, select_pred_pairs


class RunningMeanStd:
    """util_numerical.py:278-350 on the device (HIP reduction kernel)."""

    def __init__(self, name="", epsilon=1e-4, shape=(), dtype="torch.float32",
                 device="cpu"):
        self.name = "running_mean_std" if name == "" else name
        self.shape = shape
        self.dtype, self.device = util.parse_dtype_device(dtype, device)
        self.mean = torch.zeros(shape, dtype=self.dtype, device=self.device)
        self.var = torch.ones(shape, dtype=self.dtype, device=self.device)
        self.count = epsilon

    def update(self, arr):
        """RunningMeanStd.update.  With the envs sharded over ranks the batch
        statistics are those of the GLOBAL batch (pooled over the ranks before
        the running merge), so every rank keeps the same normalisation and it
        equals the single-process result."""
        self._merge(lambda mean, var, count:
                    ops.rms_update(arr, mean, var, count))

    def update_from_partials(self, partials, rows):
        """The same update from column-moment partials [*, D, 2] that were
        accumulated relative to the current running mean while the batch was
        being produced (the env rollout kernel): no second pass over it."""
        shift = self.mean.clone()
        from ..dist import active, all_reduce
        if active() and getattr(self, "equal_shards", False):
            # env shards accumulate their column moments about the SAME shift
            # (the running mean is replicated), so the shards' sums simply add:
            # one reduction, one small exchange, then the single-process merge
            # over the global row count -- instead of per-rank (mean, var, n)
            # triples pooled by ~40 small device ops
            import torch.distributed as dist
            red = partials.reshape(-1, *partials.shape[-2:]).sum(0, keepdim=True)
            all_reduce(red)
            self.count = ops.rms_merge(red, rows * dist.get_world_size(),
                                       self.mean, self.var, self.count,
                                       shift=shift)
            return
        self._merge(lambda mean, var, count:
                    ops.rms_merge(partials, rows, mean, var, count,
                                  shift=shift))

    def _merge(self, fold):
        """fold(mean, var, count) -> new count merges the local batch into the
        given running state in place."""
        import torch.distributed as dist
        from ..dist import active, all_gather_into_tensor
        if not active():
            self.count = fold(self.mean, self.var, self.count)
            return
        # local batch moments through the same kernels (merge into an empty state)
        bm = torch.zeros_like(self.mean)
        bv = torch.zeros_like(self.var)
        n_loc = fold(bm, bv, 0.0)
        D = bm.numel()
        w = dist.get_world_size()
        mine = torch.cat([bm.double().reshape(-1), bv.double().reshape(-1),
                          bm.new_full((1,), float(n_loc)).double()])
        allr = mine.new_empty(w, 2 * D + 1)
        all_gather_into_tensor(allr, mine[None])
        n = allr[:, -1:]                                   # [w, 1]
        m, v = allr[:, :D], allr[:, D:2 * D]
        n_g = n.sum()
        mean_g = (n * m).sum(0) / n_g
        m2_g = ((n - 1) * v + n * (m - mean_g) ** 2).sum(0)
        var_g = m2_g / (n_g - 1)                           # unbiased, global batch
        # update_from_moments (util_numerical.py:322-337)
        mean, var = self.mean.double().reshape(-1), self.var.double().reshape(-1)
        delta = mean_g - mean
        tot = self.count + n_g
        new_mean = mean + delta * n_g / tot
        M2 = var * self.count + var_g * n_g + delta ** 2 * self.count * n_g / tot
        self.mean.copy_(new_mean.reshape(self.mean.shape))
        self.var.copy_((M2 / tot).reshape(self.var.shape))
        if getattr(self, "equal_shards", False):
            # (MPExperiment's shards are equal -- it checks the divisibility --
            # so the global row count needs no host read of the gathered counts:
            # that read made the host wait for the whole previous iteration in
            # every rollout of a sharded run)
            self.count = self.count + float(n_loc) * w
        else:
            self.count = self.count + float(sum(float(c) for c in
                                                allr[:, -1].tolist()))

    def copy(self):
        """util_numerical.py:296-305."""
        new = RunningMeanStd(name=self.name, shape=tuple(self.mean.shape),
                             dtype=self.dtype, device=self.device)
        new.mean, new.var = self.mean.clone(), self.var.clone()
        new.count = float(self.count)
        return new

    def combine(self, other):
        """util_numerical.py:307-313."""
        self.update_from_moments(other.mean, other.var, other.count)

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        """util_numerical.py:322-337 (the parallel-variance merge; a few
        D-element device ops -- the per-rollout update takes the fused kernels
        above)."""
        batch_mean = torch.as_tensor(batch_mean, dtype=self.dtype,
                                     device=self.device)
        batch_var = torch.as_tensor(batch_var, dtype=self.dtype,
                                    device=self.device)
        delta = batch_mean - self.mean
        tot = self.count + batch_count
        m2 = self.var * self.count + batch_var * batch_count + \
            torch.square(delta) * self.count * batch_count / tot
        self.mean = self.mean + delta * batch_count / tot
        self.var = m2 / tot
        self.count = tot

    def save(self, log_dir, epoch):
        path = util.get_training_state_save_path(log_dir, self.name, epoch)
        with open(path, "wb") as f:
            torch.save({"mean": self.mean, "var": self.var,
                        "count": self.count}, f)

    def load(self, log_dir, epoch):
        path = util.get_training_state_save_path(log_dir, self.name, epoch)
        d = torch.load(path, map_location=self.device)
        self.mean, self.var, self.count = d["mean"], d["var"], d["count"]


class AbstractSampler(ABC):
    def __init__(self):
        self.train_envs = None
        self.test_envs = None
        self.debug_env = None

    @abstractmethod
    def run(self, *args, **kwargs):
        pass

    @property
    def observation_space(self):
        return self.debug_env.observation_space

    @property
    def observation_shape(self):
        return self.debug_env.observation_space.shape

    @property
    def action_space(self):
        return self.debug_env.action_space

    @property
    def spec(self):
        return self.debug_env.spec


class BlackBoxSampler(AbstractSampler):
    black_box_env = True

    def __init__(self, env_id, num_env_train=1, num_env_test=1,
                 episodes_per_train_env=1, episodes_per_test_env=1,
                 dtype="torch.float32", device="cpu", seed=1, **kwargs):
        super().__init__()
        self.env_id = env_id
        self.num_env_train, self.num_env_test = num_env_train, num_env_test
        self.episodes_per_train_env = episodes_per_train_env
        self.episodes_per_test_env = episodes_per_test_env
        self.mp_args = kwargs["mp"]["args"] if kwargs.get("mp") is not None \
            else dict()
        self.dtype, self.device = util.parse_dtype_device(dtype, device)
        self.seed = seed
        self.cpu_cores = kwargs.get("cpu_cores", None)
        self.task_specified_metrics = kwargs.get("task_specified_metrics", None)
        self.render_test_env = kwargs.get("render_test_env", False)
        self.env_args = kwargs.get("env_args", {}) or {}
        # "synthetic": the GPU-resident env suite (envs/synthetic.py);
        # "vec": any SB3-style vec env speaking the reference's protocol
        # (list of per-env numpy info dicts) behind envs/vec_adapter.py --
        # vec_env_fn(env_id=, num_env=, seed=, render=, mp_args=) builds it
        # (a callable or "module:function"; default: the reference's own
        # make_bb_vec_env over fancy_gym, mprl/util/util_mp.py:144-185)
        self.env_backend = kwargs.get("env_backend", "synthetic")
        self.vec_env_fn = kwargs.get("vec_env_fn", None)
        if self.env_backend not in ("synthetic", "vec"):
            raise ValueError("env_backend %r (synthetic | vec)"
                             % (self.env_backend,))
        self.train_envs = self.get_env("training")
        self.test_envs = self.get_env("testing")
        self.debug_env = self.get_env("debugging")

    def get_env(self, env_type="training"):
        if env_type == "training":
            num_env, seed = self.num_env_train, self.seed
        elif env_type == "testing":
            num_env, seed = self.num_env_test, self.seed + 10000
        elif env_type == "debugging":
            num_env, seed = 1, self.seed + 20000
        else:
            raise ValueError("Unknown env_type: {}".format(env_type))
        if self.env_backend == "vec":
            from ..envs import vec_adapter
            fn = vec_adapter.resolve_callable(
                self.vec_env_fn or vec_adapter.make_bb_vec_env)
            vec = fn(env_id=self.env_id, num_env=num_env, seed=seed,
                     render=self.render_test_env and env_type == "testing",
                     mp_args=self.mp_args, **self.env_args)
            return vec_adapter.VecEnvAdapter(
                vec, dtype=self.dtype, device=self.device,
                last_element_keys=self.task_specified_metrics)
        return make_env(self.env_id, num_env, seed, mp_args=self.mp_args,
                        black_box=self.black_box_env, dtype=self.dtype,
                        device=self.device, **self.env_args)

    @torch.no_grad()
    def run(self, training, policy, critic, deterministic=False, render=False,
            task_specified_metrics=None):
        if training:
            assert deterministic is False
            envs, num_env = self.train_envs, self.num_env_train
            ep_per_env = self.episodes_per_train_env
        else:
            envs, num_env = self.test_envs, self.num_env_test
            ep_per_env = self.episodes_per_test_env
        state = envs.reset()
        dim_mp_params = policy.dim_out
        out = {k: [] for k in ("segment_state", "segment_action",
                               "segment_reward", "segment_value",
                               "segment_done", "segment_log_prob",
                               "segment_params_mean", "segment_params_L")}
        metrics = {m: [] for m in (self.task_specified_metrics or [])}
        num_steps = 0
        for _ in range(ep_per_env):
            mean, L = policy.policy(state)
            assert_shape(mean, [num_env, dim_mp_params])
            action = policy.sample(require_grad=False, params_mean=mean,
                                   params_L=L, use_mean=deterministic)
            log_prob = policy.log_prob(action, params_mean=mean, params_L=L)
            values = critic.critic(state).squeeze(-1)
            out["segment_state"].append(state)
            out["segment_action"].append(action)
            out["segment_log_prob"].append(log_prob)
            out["segment_value"].append(values)
            out["segment_params_mean"].append(mean)
            out["segment_params_L"].append(L)
            state, reward, done, info = envs.step(action)
            out["segment_reward"].append(reward.to(self.dtype))
            out["segment_done"].append(done)
            # (extension: an env that knows its step count on the host says so,
            # and the rollout ends without a device -> host read)
            num_steps += info.get("num_steps_host",
                                  None) or info["trajectory_length"].sum()
            for m in metrics:
                metrics[m].append(_last_element(info[m]).to(self.dtype))
        res = {}
        for k, v in out.items():
            res[k] = _cat_L(v) if k == "segment_params_L" else torch.cat(v, 0)
        res["episode_reward"] = res["segment_reward"]
        for m, v in metrics.items():
            res[m] = torch.cat(v, 0)
        return res, int(num_steps)


def _last_element(v):
    """A task metric as the reference logs it: the LAST element of a per-step
    sequence (``get_item_from_dicts(infos, metric, lambda x: x[-1])``,
    mprl/rl/sampler/temporal_correlated_sampler.py:309-310); envs and adapters
    that reduced it already hand over [N]."""
    return v[..., -1] if v.dim() > 1 else v


def _cat_L(Ls):
    """Concatenate Cholesky factors over episodes, keeping the shared
    (stride-0) representation when every episode used the same matrix."""
    if len(Ls) == 1:
        return Ls[0]
    bases = [getattr(L, "_tce_base", None) for L in Ls]
    if all(b is not None for b in bases) and \
            all(b is bases[0] or torch.equal(b, bases[0]) for b in bases):
        return ops.expand_shared(bases[0], sum(L.shape[0] for L in Ls))
    return torch.cat([ops.full_L(L, L.shape[0]) for L in Ls], 0)


class TemporalCorrelatedSampler(BlackBoxSampler):
    black_box_env = False

    def __init__(self, env_id, num_env_train=1, num_env_test=1,
                 episodes_per_train_env=1, episodes_per_test_env=1,
                 dtype="torch.float32", device="cpu", seed=1, **kwargs):
        super().__init__(env_id, num_env_train, num_env_test,
                         episodes_per_train_env, episodes_per_test_env, dtype,
                         device, seed, **kwargs)
        self.dt = self.debug_env.envs[0].dt
        self.num_times = self.debug_env.envs[0].spec.max_episode_steps
        self.norm_step_obs = kwargs.get("norm_step_obs", False)
        self.obs_rms = RunningMeanStd(
            name="obs_rms", shape=self.observation_space.shape, dtype=dtype,
            device=device) if self.norm_step_obs else None
        self.norm_step_rewards = kwargs.get("norm_step_rewards", False)
        self.rwd_rms = RunningMeanStd(name="rwd_rms", shape=(1,), dtype=dtype,
                                      device=device) \
            if self.norm_step_rewards else None
        self.time_pairs_config = kwargs["time_pairs_config"]
        self.pred_pairs = None

    def get_times(self, init_time, num_times):
        return ops.times(init_time, self.dt, num_times)

    def get_time_pairs(self):
        """Host draw (global torch CPU generator, bit-identical to the
        reference), float32 -> int64, then one small upload."""
        pairs = select_pred_pairs(num_all=self.num_times,
                                  **self.time_pairs_config)
        pairs = pairs.to(torch.long)
        if self.device.type == "cuda":
            # from pinned memory, without waiting for the stream: an upload from
            # pageable memory blocks the host until everything enqueued before
            # it has run (the previous iteration's critic epochs, see
            # agent.lazy_metrics).  A ring of buffers, each with the event of its
            # last upload: a buffer is rewritten only after that copy has run,
            # however far the host is ahead (the lazy step keeps it within two
            # iterations, so the wait is normally over long before).
            ring = self.__dict__.setdefault("_pairs_pinned", [])
            if len(ring) < 3 or ring[0][0].shape != pairs.shape:
                ring[:] = [[torch.empty(pairs.shape, dtype=torch.long).pin_memory(),
                            None] for _ in range(3)]
                self._pairs_turn = 0
            slot = ring[self._pairs_turn % len(ring)]
            self._pairs_turn += 1
            if slot[1] is not None:
                slot[1].synchronize()
            slot[0].copy_(pairs)
            self.pred_pairs = slot[0].to(self.device, non_blocking=True)
            slot[1] = torch.cuda.Event()
            slot[1].record()
        else:
            self.pred_pairs = pairs.to(self.device)
        # env shards of one job use the SAME segments: rank 0's draw
        from ..dist import active, broadcast
        if active():
            broadcast(self.pred_pairs, src=0)
        return self.pred_pairs

    @staticmethod
    def apply_normalization(raw, rms, inplace=False):
        return ops.rms_normalize(raw, rms.mean, rms.var, 1e-8, inplace=inplace)

    @torch.no_grad()
    def run(self, training, policy, critic, deterministic=False, render=False,
            task_specified_metrics=None):
        if training:
            assert deterministic is False and render is False
            envs, num_env = self.train_envs, self.num_env_train
            ep_per_env = self.episodes_per_train_env
        else:
            envs, num_env = self.test_envs, self.num_env_test
            ep_per_env = self.episodes_per_test_env
        init_state = envs.reset()
        dim_obs = self.observation_space.shape[-1]
        num_times, num_dof = self.num_times, policy.num_dof
        pred_pairs = self.get_time_pairs()
        keys = ("step_actions", "segment_log_prob_estimate", "step_states",
                "step_rewards", "segment_state", "episode_reward",
                "step_dones", "step_values", "segment_init_time",
                "segment_init_pos", "segment_init_vel", "segment_params_mean",
                "segment_params_L")
        out = {k: [] for k in keys}
        metrics = {m: [] for m in (self.task_specified_metrics or [])}
        num_steps = 0
        for _ in range(ep_per_env):
            init_time = init_state[..., -num_dof * 2 - 1]
            init_pos = init_state[..., -num_dof * 2: -num_dof]
            init_vel = init_state[..., -num_dof:]
            mean, L = policy.policy(init_state[..., :-num_dof * 2])
            step_times = self.get_times(init_time, num_times)
            actions = policy.sample(require_grad=False, params_mean=mean,
                                    params_L=L, times=step_times,
                                    init_time=init_time, init_pos=init_pos,
                                    init_vel=init_vel, use_mean=deterministic)
            log_prob = policy.log_prob(actions, params_mean=mean, params_L=L,
                                       times=step_times, init_time=init_time,
                                       init_pos=init_pos, init_vel=init_vel,
                                       pred_pairs=pred_pairs)
            assert_shape(actions, [num_env, num_times, num_dof * 2])
            # only updated AND applied during training: evaluation feeds the
            # critic raw states (temporal_correlated_sampler.py:244-249)
            norm = self.norm_step_obs and training
            fused_env = getattr(envs, "fused_obs_moments", False)
            if norm and fused_env:
                # the env kernel writes the [N, T+1, D] buffer (initial state
                # in row 0) once and sums the column moments in the same pass
                next_state, ep_reward, _, infos = envs.step(
                    actions, obs_shift=self.obs_rms.mean, want_moments=True)
            else:
                next_state, ep_reward, _, infos = envs.step(actions)
            if "step_states_full" in infos:
                step_states = infos["step_states_full"]
                own_buffer = fused_env
            else:
                # an env that speaks the reference protocol only: step_states
                # [N, T, D], the initial state prepended here
                # (temporal_correlated_sampler.py:233-240)
                step_states = torch.cat(
                    [init_state[:, None].to(self.dtype),
                     infos["step_states"].to(self.dtype)], dim=1)
                own_buffer = True
            if norm and fused_env:
                self.obs_rms.update_from_partials(
                    infos["obs_moment_partials"],
                    rows=step_states.shape[0] * step_states.shape[1])
                norm_states = self.apply_normalization(
                    step_states, self.obs_rms, inplace=True)
            elif norm:
                self.obs_rms.update(step_states.reshape(-1, dim_obs))
                norm_states = self.apply_normalization(
                    step_states, self.obs_rms, inplace=own_buffer)
            else:
                norm_states = step_states
            values = critic.critic(
                norm_states[..., :-num_dof * 2]).squeeze(-1)
            rewards = infos["step_rewards"].to(self.dtype)
            if "TableTennis" in self.env_id:
                rewards = ops.mdp_reward(rewards, infos["hit_ball"])
            elif "HopperJump" in self.env_id:
                rewards = ops.mdp_reward(rewards, infos["has_left_floor"])
            dones = torch.logical_or(infos["step_terminations"],
                                     infos["step_truncations"])
            out["step_actions"].append(actions)
            out["segment_log_prob_estimate"].append(log_prob)
            out["step_states"].append(norm_states)
            out["step_values"].append(values)
            out["segment_state"].append(init_state)
            out["step_rewards"].append(rewards)
            out["episode_reward"].append(ep_reward.to(self.dtype))
            out["step_dones"].append(dones)
            out["segment_init_time"].append(init_time)
            out["segment_init_pos"].append(init_pos)
            out["segment_init_vel"].append(init_vel)
            out["segment_params_mean"].append(mean)
            out["segment_params_L"].append(L)
            num_steps = num_steps + (infos.get("num_steps_host", None) or
                                     infos["segment_length"].sum())
            for m in metrics:
                metrics[m].append(_last_element(infos[m]).to(self.dtype))
            init_state = next_state
        res = {}
        for k, v in out.items():
            if k == "segment_params_L":
                res[k] = _cat_L(v)
            else:
                res[k] = v[0] if len(v) == 1 else torch.cat(v, dim=0)
        # the reference stores the normalised states incl. the initial one and
        # drops the last (temporal_correlated_sampler.py:322); kept as a view
        res["step_states_full"] = res["step_states"]
        res["step_states"] = res["step_states"][:, :-1]
        res["segment_reward"] = res["step_rewards"].sum(dim=-1)
        res["step_time_limit_dones"] = torch.zeros_like(res["step_dones"])
        for m, v in metrics.items():
            res[m] = torch.cat(v, dim=0)
        return res, int(num_steps)

    def save_rms(self, log_dir, epoch):
        if self.norm_step_obs:
            self.obs_rms.save(log_dir, epoch)
        if self.norm_step_rewards:
            self.rwd_rms.save(log_dir, epoch)

    def load_rms(self, log_dir, epoch):
        if self.norm_step_obs:
            self.obs_rms.load(log_dir, epoch)
        if self.norm_step_rewards:
            self.rwd_rms.load(log_dir, epoch)


def sampler_factory(typ, **kwargs):
    return {"BlackBoxSampler": BlackBoxSampler,
            "TemporalCorrelatedSampler": TemporalCorrelatedSampler}[typ](
        **kwargs)
