"""Agents: mirror of mprl/rl/agent/ (abstract_agent.py:12-255,
temporal_correlated_agent.py:12-753, black_box_agent.py:12-495).

``step()`` keeps the whole iteration on the device: rollout buffers, GAE /
segment advantages, critic and policy epochs.  Per-epoch scalars are collected
in device tensors and copied to the host ONCE at the end of the update (the
reference synchronises >= 20 times per policy epoch: ``.item()`` x4,
``to_np`` x12, NaN checks x3, grad-norm ``.item()`` per parameter).

Multi-GPU (env sharding): every rank holds ``num_env_train / world`` envs and a
replica of policy / critic; gradients are summed with one flat all-reduce per
optimizer step and divided by the world size (local losses are means over the
local shard, shards are equal, so this equals the reference's global mean);
advantage statistics are merged over ranks before normalisation.
"""
import os
from abc import ABC, abstractmethod

import numpy as np
import torch
from torch.optim.lr_scheduler import LinearLR

from .. import ops, util
from ..dist import DistContext
from ..optim import FlatAdam
from . import objective
from .projection import gaussian_kl_details


class AbstractAgent(ABC):
    def __init__(self, policy, critic, sampler, projection,
                 dtype="torch.float32", device="cpu", **kwargs):
        self.policy, self.critic = policy, critic
        self.sampler, self.projection = sampler, projection
        self.dtype, self.device = util.parse_dtype_device(dtype, device)
        self.lr_policy = float(kwargs["lr_policy"])
        self.lr_critic = float(kwargs["lr_critic"])
        self.wd_policy = float(kwargs["wd_policy"])
        self.wd_critic = float(kwargs["wd_critic"])
        self.schedule_lr_policy = kwargs.get("schedule_lr_policy", False)
        self.schedule_lr_critic = kwargs.get("schedule_lr_critic", False)
        self.total_iterations = kwargs.get("total_iterations", 10000)
        self.discount_factor = float(kwargs["discount_factor"])
        self.epochs_policy = kwargs["epochs_policy"]
        self.epochs_critic = kwargs["epochs_critic"]
        self.dist = DistContext(kwargs.get("process_group", None))
        self.policy_net_params = None
        self.critic_net_params = None
        self.policy_optimizer, self.critic_optimizer = \
            self.get_optimizer(self.policy, self.critic)
        self.policy_lr_scheduler, self.critic_lr_scheduler = \
            self.get_lr_scheduler()
        self.num_iterations = 0
        self.num_global_steps = 0
        self._policy_group = None
        # the gradient exchanges of the two update chains (env shards): inside
        # the library (dist.Exchange, one-shot over xGMI) or -- None -- as
        # torch.distributed all-reduces between the C calls
        self.xchg_critic = self.xchg_policy = None
        if self.dist.active:
            nbytes = lambda opt: (opt.flat_grad.numel() + 64) * \
                opt.flat_grad.element_size()
            self.xchg_critic = self.dist.exchange(
                "critic", nbytes(self.critic_optimizer))
            self.xchg_policy = self.dist.exchange(
                "policy", nbytes(self.policy_optimizer))
            # the small per-step collectives (statistics, pairs, critic split)
            # ride on a third exchange: no torch.distributed call in a step
            self.dist.setup_aux()
            if self.xchg_policy is None:
                # (torch.distributed path: the policy's all-reduces need their
                # own communicator, see DistContext.aux_group)
                self._policy_group = self.dist.aux_group()
            self.dist.broadcast_params(self.policy_net_params +
                                       self.critic_net_params)

    def get_optimizer(self, policy, critic):
        """Adam with L2-in-gradient weight decay (abstract_agent.py:62-82)."""
        self.policy_net_params = policy.parameters
        self.critic_net_params = critic.parameters
        mk = lambda params, lr, wd: FlatAdam(params, lr=lr, weight_decay=wd)
        return mk(self.policy_net_params, self.lr_policy, self.wd_policy), \
            mk(self.critic_net_params, self.lr_critic, self.wd_critic)

    def get_lr_scheduler(self):
        mk = lambda opt: LinearLR(opt, start_factor=1, end_factor=0.01,
                                  total_iters=self.total_iterations)
        return (mk(self.policy_optimizer) if self.schedule_lr_policy else None,
                mk(self.critic_optimizer) if self.schedule_lr_critic else None)

    def save_agent(self, log_dir, epoch):
        if hasattr(self, "flush_metrics"):
            self.flush_metrics()        # deferred NaN checks before a checkpoint
        self.policy.save_weights(log_dir, epoch)
        self.critic.save_weights(log_dir, epoch)
        for name, opt in (("policy_optimizer", self.policy_optimizer),
                          ("critic_optimizer", self.critic_optimizer)):
            path = util.get_training_state_save_path(log_dir, name, epoch)
            with open(path, "wb") as f:
                torch.save(opt.state_dict(), f)

    def load_agent(self, log_dir, epoch):
        self.policy.load_weights(log_dir, epoch)
        self.critic.load_weights(log_dir, epoch)
        self.policy_optimizer, self.critic_optimizer = \
            self.get_optimizer(self.policy, self.critic)
        for name, opt in (("policy_optimizer", self.policy_optimizer),
                          ("critic_optimizer", self.critic_optimizer)):
            path = util.get_training_state_save_path(log_dir, name, epoch)
            opt.load_state_dict(torch.load(path, map_location=self.device))
        self.policy_lr_scheduler, self.critic_lr_scheduler = \
            self.get_lr_scheduler()
        # epoch None = the un-suffixed files (util_file.py:293-317); the
        # reference then leaves num_iterations = None (abstract_agent.py:174),
        # which only an evaluation run survives -- count from 0 instead
        self.num_iterations = 0 if epoch is None else epoch

    @abstractmethod
    def step(self, *args, **kwargs):
        pass

    @abstractmethod
    def update_policy(self, *args, **kwargs):
        pass

    @abstractmethod
    def update_critic(self, *args, **kwargs):
        pass

    @torch.no_grad()
    def evaluate(self, evaluate_deterministic=True, evaluate_stochastic=False,
                 render=False):
        det = self.sampler.run(training=False, policy=self.policy,
                               critic=self.critic,
                               deterministic=evaluate_deterministic,
                               render=render)[0] \
            if evaluate_deterministic else dict()
        sto = self.sampler.run(training=False, policy=self.policy,
                               critic=self.critic, deterministic=False,
                               render=render)[0] \
            if evaluate_stochastic else dict()
        return det, sto

    # ---- shared pieces of the update loops --------------------------------
    def _grad_norm_clip(self, bound, params):
        """util_numerical.py:244-275 without host syncs: returns the two norms
        as 0-dim device tensors."""
        grads = [p.grad for p in params]
        flat = torch.cat([g.reshape(-1) for g in grads])
        before = flat.norm(2)
        if bound > 0:
            coef = torch.clamp(bound / (before + 1e-6), max=1.0)
            for g in grads:
                g.mul_(coef)
            after = before * coef
        else:
            after = before
        return before, after

    def _optimizer_step(self, opt, params, clip, want_norms=True):
        """grad_norm_clip + Adam step (one flat buffer; one collective when
        the envs are sharded over ranks)."""
        if self.dist.active:
            opt.sync_grads()
            policy = opt is self.policy_optimizer
            xch = self.xchg_policy if policy else self.xchg_critic
            if xch is not None and opt.flat_grad.numel() <= (1 << 17):
                # sum over the shards + clip + Adam: one C call (tce_xchg_adam_*)
                opt.step_exchange(xch, clip, grad_scale=1.0 / self.dist.world)
                if not want_norms:
                    return None                 # the caller reads dev_state[1:3]
                norms = opt.dev_state[1:3].clone()
                return norms[0], norms[1]
            if xch is not None:
                xch.allreduce(opt.flat_grad)
            else:
                # the policy's exchange has its own communicator (see dist.py)
                self.dist.allreduce_flat(
                    opt.flat_grad, self._policy_group if policy else None,
                    average=False)
        if self.dist.active and not want_norms:
            opt.step_once(clip, grad_scale=1.0 / self.dist.world)
            return None                         # the caller reads dev_state[1:3]
        opt.step(clip, grad_scale=1.0 / self.dist.world)   # mean over ranks
        if not want_norms:                      # the caller reads dev_state[1:3]
            return None
        norms = opt.dev_state[1:3].clone()      # the state is reused next step
        return norms[0], norms[1]

    def _capture(self, fn, pool_key=None):
        """Record fn() (kernel launches only, fixed buffers) into a HIP graph
        on a side stream, without the device-wide synchronisation of
        torch.cuda.graph() -- the critic epochs keep running meanwhile.
        pool_key: graphs that are replayed CONCURRENTLY (the black-box agent's
        critic and policy epochs) must not share a memory pool."""
        if getattr(self, "_graph_stream", None) is None:
            from .. import streams
            self._graph_stream = streams.get("graph", self.device)
            self._graph_pools = {}
        if pool_key not in self._graph_pools:
            self._graph_pools[pool_key] = torch.cuda.graph_pool_handle()
        graph = torch.cuda.CUDAGraph()
        cur = torch.cuda.current_stream()
        self._graph_stream.wait_stream(cur)
        with torch.cuda.stream(self._graph_stream):
            graph.capture_begin(pool=self._graph_pools[pool_key])
            try:
                fn()
            finally:
                graph.capture_end()
        cur.wait_stream(self._graph_stream)
        return graph

    def _run_epochs(self, epoch, E, opt, graph):
        """E identical epochs (fixed buffers, no host reads).  graph: the first
        epoch runs eagerly, the second is recorded into a HIP graph and replayed
        -- the ~100 launches of an epoch leave the host."""
        if graph and E > 2 and not self.dist.active:
            # the very first update runs one epoch eagerly (lazy initialisation
            # of the GEMM library must not happen under capture); afterwards
            # all E epochs are replays -- an eager epoch costs 1.5 - 4 ms of
            # host time and these updates are host-bound
            n = E
            if not getattr(opt, "_tce_graph_warm", False):
                epoch()
                opt._tce_graph_warm = True
                n = E - 1
            g = self._capture(epoch, pool_key=id(opt))
            for _ in range(n):
                g.replay()
            opt.host_step += n - 1                # the capture counted one
            self._last_graphs = getattr(self, "_last_graphs", [])[-3:] + [g]
        else:
            for _ in range(E):
                epoch()

    def _nan_over_ranks(self, losses):
        """losses [E, 3] (surrogate, entropy, trust-region loss per epoch, on
        the device) -> [3] float64 device tensor, 1 where a loss was NaN in any
        epoch ON ANY RANK.  Env shards: the flags ride on the small-collective
        exchange (one launch, enqueued in step with the peers; no host wait),
        so the check of temporal_correlated_agent.py:569-577 raises on every
        rank in the same iteration -- not on the rank with the bad shard alone
        while its peers run into their next bounded wait."""
        f = torch.isnan(losses).any(dim=0).to(torch.float64)
        if self.dist.active:
            import torch.distributed as tdist
            from ..dist import all_reduce
            all_reduce(f, op=tdist.ReduceOp.MAX,
                       group=self._policy_group or self.dist.group)
        return f

    @staticmethod
    def _raise_on_nan(flags_host):
        for name, bad in zip(("surrogate_loss", "entropy_loss",
                              "trust_region_loss"), flags_host):
            if bad:
                raise Exception("NAN %s detected" % name)

    def _critic_minibatches_fused(self):
        """Can ``num_minibatchs`` optimizer steps per epoch run inside the
        matrix-core critic epochs (tce_mlp_critic_minibatch_f32 /
        tce_mlpw_critic_minibatch_*: gathered rows, one C call per epoch)?
        Always for one minibatch; else whenever the gradient needs no
        torch.distributed all-reduce between the kernels."""
        k = int(getattr(self, "num_minibatchs", 1) or 1)
        return k == 1 or not self.dist.active or self.xchg_critic is not None

    def close(self):
        """Env shards: release the peer-visible exchange buffers (collective --
        every rank calls it at the same point, before the process group is
        destroyed).  A no-op for a single-process agent."""
        flush = getattr(self, "flush_metrics", None)
        if flush is not None:
            flush()
        self.dist.close()
        self.xchg_critic = self.xchg_policy = None


class _CriticEpochs:
    """Full-batch critic epochs on the fused fp32-MFMA kernel: one launch does
    forward + value loss + backward for all N*T rows (read in place from the
    rollout buffer), a second reduces the per-workgroup gradient slabs and
    applies Adam.  ``run`` may be called in pieces with different workgroup
    limits (the overlapped update gives the critic the whole chip once the
    policy epochs are done)."""

    def __init__(self, agent, x, returns, old_values):
        from .. import critic_ops
        self.agent = agent
        self.x, self.returns, self.old_values = x, returns, old_values
        opt = self.opt = agent.critic_optimizer
        run = getattr(agent, "_critic_runner", None)
        arith = getattr(agent, "critic_arith", "f32")
        if critic_ops.wide_supported(agent.critic.net) or \
                int(getattr(agent, "num_minibatchs", 1) or 1) > 1:
            # exact matrix cores of the net's own dtype (the split-operand
            # kernels have no gathered-row form)
            arith = "f32"
        if run is None or run.mlp is not agent.critic.net or \
                run.flat is not opt.flat_grad or run.arith != arith:
            run = agent._critic_runner = critic_ops.make_runner(
                agent.critic.net, opt.flat_grad, arith=arith)
        self.runner = run
        opt.bind_grads()
        self.E = agent.epochs_critic
        # minibatches (the reference's class default is 10,
        # temporal_correlated_agent.py:25,343-366): an epoch is ONE C call that
        # takes `k` optimizer steps over gathered rows
        self.k = int(getattr(agent, "num_minibatchs", 1) or 1)
        self.n_rows = int(returns.numel())
        self._perm_ring = []
        # per optimizer step {mean loss, |g|^2 (accumulated by the kernel), |g|,
        # |g| clipped}
        self.rows = torch.zeros(self.E * self.k, 4,
                                dtype=agent.critic.net.dtype,
                                device=agent.device)
        # env shards: the exchange rides in the launch that applies Adam
        self.xchg = agent.xchg_critic if agent.dist.active else None
        self.gscale = 1.0 / agent.dist.world if agent.dist.active else 1.0
        self.fuse_adam = (not agent.dist.active or self.xchg is not None) \
            and not agent.clip_grad_norm > 0
        self.done = 0

    def _permutation(self):
        """The epoch's row permutation on the device.  "numpy" (default): the
        reference's own draw -- np.random.shuffle of arange(n) on numpy's GLOBAL
        generator (generate_minibatches, util_data_structure.py:378-391), i.e.
        the same minibatches as the reference from the same seed; a sequential
        Fisher-Yates on the host (~13 ns per row), uploaded through one of two
        pinned buffers while the previous epoch runs.  "device":
        torch.randperm on the GPU (the device generator) -- statistically the
        same, not the reference's sequence; for runs where the host draw
        (28 ms per epoch at 2 M rows) would be the step."""
        ag, n = self.agent, self.n_rows
        if getattr(ag, "minibatch_permutation", "numpy") == "device":
            return torch.randperm(n, device=ag.device, dtype=torch.int64)
        idx = np.arange(n)
        np.random.shuffle(idx)
        if len(self._perm_ring) < 2:
            host = torch.empty(n, dtype=torch.int64).pin_memory()
            self._perm_ring.append([host, None])
        slot = self._perm_ring[0]
        self._perm_ring.reverse()
        if slot[1] is not None:
            slot[1].synchronize()         # its previous upload has left the host
        slot[0].numpy()[:] = idx
        dev = slot[0].to(ag.device, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record()
        return dev

    def _run_minibatched(self, n, max_workgroups):
        ag, opt, k = self.agent, self.opt, self.k
        for e in range(self.done, min(self.E, self.done + n)):
            self.runner.epoch_minibatches(
                self.x, self.returns, self.old_values, ag.clip_critic,
                self._permutation(), k, self.rows[e * k:(e + 1) * k], opt,
                grad_clip=ag.clip_grad_norm, max_workgroups=max_workgroups,
                xchg=self.xchg, grad_scale=self.gscale)
            self.done = e + 1

    def run(self, n, max_workgroups=0):
        if self.k > 1:
            return self._run_minibatched(n, max_workgroups)
        ag, opt, rows = self.agent, self.opt, self.rows
        for e in range(self.done, min(self.E, self.done + n)):
            fused = self.fuse_adam
            self.runner.epoch(self.x, self.returns, self.old_values,
                              ag.clip_critic, max_workgroups, stats=rows[e],
                              adam=opt if fused else None,
                              xchg=self.xchg if fused else None,
                              grad_scale=self.gscale if fused else 1.0)
            if not self.fuse_adam:
                if ag.dist.active and self.xchg is not None and \
                        opt.flat_grad.numel() <= (1 << 17):
                    # sum over the shards + clip + Adam + the record's norms:
                    # one C call (tce_xchg_adam_*)
                    opt.step_exchange(self.xchg, ag.clip_grad_norm,
                                      grad_scale=self.gscale,
                                      norms_out=rows[e, 2:4])
                elif ag.dist.active:
                    # sum over the shards, then clip + Adam + the two norms of
                    # the record in ONE launch (tce_adam_once_*)
                    if self.xchg is not None:
                        self.xchg.allreduce(opt.flat_grad)
                    else:
                        ag.dist.allreduce_flat(opt.flat_grad, average=False)
                    opt.step_once(ag.clip_grad_norm,
                                  grad_scale=self.gscale,
                                  norms_out=rows[e, 2:4])
                else:                   # |g|^2 comes with the reduction
                    opt.step(ag.clip_grad_norm, sumsq=rows[e, 1:2])
                    rows[e, 2:4].copy_(opt.dev_state[1:3])
            self.done = e + 1

    def finish(self):
        host = self.rows.cpu().numpy()                       # the only sync
        if self.runner.arith == "f16x2" and not np.isfinite(host[:, 0]).all():
            raise RuntimeError(
                "critic_arith=f16x2: the critic loss is not finite -- an "
                "operand (observation, activation, weight) left the f16 range "
                "(|x| < 65504); use critic_arith=f32 for this task")
        if self.fuse_adam and self.xchg is None:             # no clipping
            # (env shards: the exchange's Adam launch has written both norms)
            host[:, 2] = host[:, 3] = np.sqrt(host[:, 1])
        return {**util.generate_stats(host[:, 0], "critic_loss"),
                **util.generate_stats(host[:, 2], "critic_grad_norm"),
                **util.generate_stats(host[:, 3], "clipped_critic_grad_norm")}


class TemporalCorrelatedAgent(AbstractAgent):
    def __init__(self, policy, critic, sampler, projection,
                 dtype=torch.float32, device=torch.device("cpu"), **kwargs):
        super().__init__(policy, critic, sampler, projection, dtype=dtype,
                         device=device, **kwargs)
        self.clip_critic = float(kwargs.get("clip_critic", 0.0))
        self.clip_grad_norm = float(kwargs.get("clip_grad_norm", 0.0))
        self.num_minibatchs = kwargs.get("num_minibatchs", 10)
        # who draws the critic's minibatch permutations: "numpy" = the
        # reference's own draw on numpy's global generator
        # (util_data_structure.py:389-390: same pieces from the same seed; a
        # sequential host shuffle), "device" = torch.randperm on the GPU
        # (statistically the same, not the reference's sequence)
        self.minibatch_permutation = kwargs.get("minibatch_permutation",
                                                "numpy")
        if self.minibatch_permutation not in ("numpy", "device"):
            raise NotImplementedError(
                "minibatch_permutation=%r (numpy | device)"
                % (self.minibatch_permutation,))
        self.norm_advantages = kwargs.get("norm_advantages", False)
        self.clip_advantages = kwargs.get("clip_advantages", False)
        self.entropy_penalty_coef = float(
            kwargs.get("entropy_penalty_coef", 0.0))
        self.use_gae = kwargs.get("use_gae", True)
        self.gae_scaling = float(kwargs.get("gae_scaling", 0.95))
        self.segment_advantage = kwargs.get("segment_advantage", "accumulate")
        self.set_variance = kwargs.get("set_variance", False)
        self.balance_check = kwargs.get("balance_check", 10)
        self.evaluation_interval = kwargs.get("evaluation_interval", 1)
        self.check_policy_balance = False
        # extension: run the critic and policy updates on two HIP streams
        self.overlap_updates = kwargs.get("overlap_updates", True)
        # hipGraph replay of the policy epochs: fewer host launches, but the
        # node-to-node latency grows ~10x while another stream keeps the GPU
        # busy (measured), so it only pays without the overlapped critic
        self.graph_policy_update = kwargs.get("graph_policy_update", False)
        self.fused_policy_objective = kwargs.get("fused_policy_objective",
                                                 True)
        # the fused objective's epoch without autograd (rl/objective.py:
        # DirectEpoch): half the launches of an epoch
        self.direct_policy_epoch = kwargs.get("direct_policy_epoch", True)
        # arithmetic of the fused critic epoch: "f32" = exact-fp32 matrix cores
        # (csrc/mlp.hip); "bf16x3" = three-part bf16 operands on the bf16
        # matrix cores (csrc/mlpb.hip: x = b0 + b1 + b2 exactly -- 24 bits,
        # fp32's range -- six partial products, fp32 accumulate: as close to
        # fp64 as the fp32 kernel, 1.4x faster); "f16x2" = split-f16 operands
        # on the f16 matrix cores (csrc/mlp16.hip: 22-bit operands inside the
        # f16 range, 2.4x faster)
        self.critic_arith = kwargs.get("critic_arith", "f32")
        if self.critic_arith not in ("f32", "f16x2", "bf16x3"):
            raise NotImplementedError("critic_arith %r" % (self.critic_arith,))
        self.critic_workgroups = int(kwargs.get(
            "critic_workgroups", os.environ.get("TCE_CRITIC_WORKGROUPS", 224)))
        self.critic_cus_per_xcd = kwargs.get("critic_cus_per_xcd", None)
        self.adaptive_critic_split = kwargs.get("adaptive_critic_split", True)
        # step() returns its metrics as util.LazyMetrics (filled on first
        # access) and does not wait for the critic epochs it has enqueued: the
        # host prepares the next rollout meanwhile (overlapped updates, one
        # process; otherwise the metrics are read before step() returns)
        # (TCE_LAZY_METRICS=0: the default of this option, for A / B runs)
        self.lazy_metrics = kwargs.get(
            "lazy_metrics", os.environ.get("TCE_LAZY_METRICS", "1") != "0")
        self._lazy_done = []            # end-of-step events of the last steps
        self._split_probes = []         # events of the last steps the critic split is taken from
        # epochs of slack on the split (a policy stream that outlasts them keeps
        # the critic's remaining epochs waiting; measured at C2: 2 -> 1 epoch of
        # slack is 0.4 ms per step, 0.5 no better)
        self._split_margin = float(os.environ.get("TCE_SPLIT_MARGIN", "1"))
        self._critic_split = 0          # 0: all epochs beside the policy
        self._critic_split_bal = 0      # the same for balance-check iterations
        self._local_split = [0, 0]      # this rank's estimates (lazy steps)
        self._split_exchanges = []      # (event, pinned result) of the MAX all-reduces in flight
        self._critic_stream = None
        self._policy_stream = None

    def _lazy_step_possible(self):
        from .. import critic_ops
        return (self.lazy_metrics and self._can_overlap()
                and self.device.type == "cuda"
                and critic_ops.supported(self.critic.net))

    def _retire_lazy_steps(self, keep):
        """Lazy steps: the host runs at most `keep` iterations ahead of the
        device.  An iteration that leaves the window has finished on the device
        (its end event is waited for -- usually long past), and its metrics are
        read HERE if the caller has not read them: that read carries the checks
        the reference runs inside update_policy / update_critic in every
        iteration (NaN losses, temporal_correlated_agent.py:569-577; the f16x2
        critic's finiteness check), so a caller that never looks at the metrics
        (MPExperiment.iterate at verbose_level 0) still stops on a NaN, two
        iterations late at most, and before the next checkpoint
        (``flush_metrics``)."""
        done = self.__dict__.setdefault("_lazy_done", [])
        while done and len(done) >= keep:
            ev, metrics = done.pop(0)
            ev.synchronize()
            # (env shards: a wait for a peer that ran into its limit is fatal)
            self.dist.check_exchanges()
            metrics.resolve()

    def flush_metrics(self):
        """Wait for every enqueued iteration and run its deferred checks."""
        self._retire_lazy_steps(0)

    def _step_lazy(self):
        """step() without a host wait at its end (see lazy_metrics): phase
        times come from HIP events, the records are read when the metrics are."""
        self.num_iterations += 1
        self._retire_lazy_steps(2)
        main = torch.cuda.current_stream()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record(main)
        dataset, num_env_interaction = self.sampler.run(
            training=True, policy=self.policy, critic=self.critic)
        self.num_global_steps += num_env_interaction * self.dist.world
        ev[1].record(main)
        dataset = self.process_dataset(dataset)
        ev[2].record(main)
        stat_items = {k: v for k, v in dataset.items()
                      if k not in ("segment_params_L", "step_states_full",
                                   "step_states", "step_actions")}
        tail = self._update_overlapped(
            dataset, lambda: util.device_stats_async(stat_items, "exploration"),
            lazy=True)
        if self.schedule_lr_critic:
            self.critic_lr_scheduler.step()
        if self.schedule_lr_policy:
            self.policy_lr_scheduler.step()
        ev[3].record(main)
        steps = self.num_global_steps
        lr_p = self.policy_lr_scheduler.get_last_lr()[0] \
            if self.schedule_lr_policy else self.lr_policy
        lr_c = self.critic_lr_scheduler.get_last_lr()[0] \
            if self.schedule_lr_critic else self.lr_critic

        def resolve():
            ev[3].synchronize()
            critic_loss_dict, policy_loss_dict, t_c, t_p, dataset_stats = tail()
            return {**dataset_stats, **critic_loss_dict, **policy_loss_dict,
                    "sampling_time": ev[0].elapsed_time(ev[1]) * 1e-3,
                    "process_dataset_time": ev[1].elapsed_time(ev[2]) * 1e-3,
                    "update_time": ev[2].elapsed_time(ev[3]) * 1e-3,
                    "update_critic_time": t_c, "update_policy_time": t_p,
                    "num_global_steps": steps, "lr_policy": lr_p,
                    "lr_critic": lr_c}
        result = util.LazyMetrics(resolve)
        self._lazy_done.append((ev[3], result))
        if self.evaluation_interval and (
                self.evaluation_interval == 1 or
                self.num_iterations % self.evaluation_interval == 1):
            util.run_time_test(lock=True, key="evaluation")
            evd = self.evaluate()[0]
            result.update(util.device_stats(
                {k: v for k, v in evd.items()
                 if k not in ("segment_params_L", "step_states_full",
                              "step_states", "step_actions")}, "evaluation"))
            result["evaluation_time"] = util.run_time_test(
                lock=False, key="evaluation")
        return result

    def step(self):
        if self._lazy_step_possible():
            return self._step_lazy()
        self.num_iterations += 1
        util.run_time_test(lock=True, key="sampling")
        dataset, num_env_interaction = self.sampler.run(
            training=True, policy=self.policy, critic=self.critic)
        self.num_global_steps += num_env_interaction * self.dist.world
        sampling_time = util.run_time_test(lock=False, key="sampling")

        util.run_time_test(lock=True, key="process_dataset")
        dataset = self.process_dataset(dataset)
        process_dataset_time = util.run_time_test(lock=False,
                                                  key="process_dataset")
        # exploration statistics: reductions are enqueued behind the policy
        # update (second stream), the host reads them after the updates
        stat_items = {k: v for k, v in dataset.items()
                      if k not in ("segment_params_L", "step_states_full",
                                   "step_states", "step_actions")}

        util.run_time_test(lock=True, key="update")
        if self._can_overlap():
            critic_loss_dict, policy_loss_dict, update_critic_time, \
                update_policy_time, dataset_stats = self._update_overlapped(
                    dataset, lambda: util.device_stats_async(
                        stat_items, "exploration"))
        else:
            dataset_stats = util.device_stats(stat_items, "exploration")
            util.run_time_test(lock=True, key="update critic")
            critic_loss_dict = self.update_critic(dataset)
            update_critic_time = util.run_time_test(lock=False,
                                                    key="update critic")
            util.run_time_test(lock=True, key="update policy")
            policy_loss_dict = self.update_policy(dataset)
            update_policy_time = util.run_time_test(lock=False,
                                                    key="update policy")
        if self.schedule_lr_critic:
            self.critic_lr_scheduler.step()
        if self.schedule_lr_policy:
            self.policy_lr_scheduler.step()
        update_time = util.run_time_test(lock=False, key="update")
        self.dist.check_exchanges()

        result_metrics = {
            **dataset_stats, **critic_loss_dict, **policy_loss_dict,
            "sampling_time": sampling_time,
            "process_dataset_time": process_dataset_time,
            "update_time": update_time,
            "update_critic_time": update_critic_time,
            "update_policy_time": update_policy_time,
            "num_global_steps": self.num_global_steps,
            "lr_policy": self.policy_lr_scheduler.get_last_lr()[0]
            if self.schedule_lr_policy else self.lr_policy,
            "lr_critic": self.critic_lr_scheduler.get_last_lr()[0]
            if self.schedule_lr_critic else self.lr_critic}

        # evaluation_interval 0 / None: never (extension; reference default 1)
        if self.evaluation_interval and (
                self.evaluation_interval == 1 or
                self.num_iterations % self.evaluation_interval == 1):
            util.run_time_test(lock=True, key="evaluation")
            ev = self.evaluate()[0]
            result_metrics.update(util.device_stats(
                {k: v for k, v in ev.items()
                 if k not in ("segment_params_L", "step_states_full",
                              "step_states", "step_actions")}, "evaluation"))
            result_metrics["evaluation_time"] = util.run_time_test(
                lock=False, key="evaluation")
        return result_metrics

    # ---- critic and policy updates side by side ------------------------------
    def _can_overlap(self):
        """Critic and policy updates side by side: both must be enqueued
        without a host read in between -- one minibatch, or minibatches inside
        the fused critic epochs (the policy update is full-batch always:
        temporal_correlated_agent.py:381-639)."""
        if not self.overlap_updates:
            return False
        if self.num_minibatchs == 1:
            return True
        from .. import critic_ops
        return self.device.type == "cuda" and \
            critic_ops.supported(self.critic.net) and \
            self._critic_minibatches_fused()

    def _update_overlapped(self, dataset, side_work=None, lazy=False):
        """The critic and policy updates of one iteration touch disjoint
        networks and only read the dataset, so they are independent.  The
        critic epochs (one persistent MFMA kernel each, 1 workgroup per CU) are
        enqueued first on the main stream with a few CUs left free (one per
        shader engine: a policy kernel's workgroups are spread over all of
        them); the policy epochs (many small latency-bound kernels) run beside
        them on a second HIP stream.  The policy finishes first: the remaining
        critic epochs wait for it and then take every CU.  The split point
        follows the device times measured in the previous iteration.  Results
        are identical to the sequential order; the per-phase times reported are
        device times (HIP events)."""
        main = torch.cuda.current_stream()
        if self._policy_stream is None:
            self._make_update_streams()
        side, cstream = self._policy_stream, self._critic_stream
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        D2 = self.policy.num_dof * 2
        from .. import critic_ops
        if not critic_ops.supported(self.critic.net):
            # library-GEMM critic (256-wide nets, fp64): ordinary kernels, the
            # two streams simply share the chip
            ev[0].record(main)
            finish_critic = self.update_critic(dataset, defer=True)
            ev[1].record(main)
            side.wait_event(ev[0])
            with torch.cuda.stream(side):
                ev[2].record(side)
                policy_loss_dict = self._update_policy_beside_critic(dataset)
                ev[3].record(side)
                finish_side = side_work() if side_work is not None else dict
            main.wait_stream(side)
            critic_loss_dict = finish_critic()
            side_result = finish_side()
            torch.cuda.synchronize()
            return critic_loss_dict, policy_loss_dict, \
                ev[0].elapsed_time(ev[1]) * 1e-3, \
                ev[2].elapsed_time(ev[3]) * 1e-3, side_result
        ce = _CriticEpochs(self, dataset["step_states"][..., :-D2],
                           dataset["step_returns"],
                           dataset["step_values"][:, :-1])
        E = ce.E
        # (lazy steps) the earlier steps whose events are complete give the
        # split; with all epochs beside the policy the critic event of a step is
        # the END of its epochs, which the host may be ahead of -- such a probe
        # stays for the next look.  Iterations with the policy balance check
        # (1 in `balance_check`) have a longer policy phase and their own split.
        bal = self._balance_iteration()
        if lazy and self._early_split_exchange():
            # (sharded runs, first iterations: see _adopt_split)
            torch.cuda.synchronize()
        waiting = []
        for probe in self._split_probes:
            pev, pn1, pE, pbal = probe
            if pev[6].query() and pev[5].query():
                if self.adaptive_critic_split and cstream is None:
                    first_ms = pev[0].elapsed_time(pev[6]) / max(min(pn1, 6), 1)
                    side_ms = pev[2].elapsed_time(pev[5])
                    split = int(min(pE, side_ms / first_ms + self._split_margin))
                    self._local_split[1 if pbal else 0] = split
            else:
                waiting.append(probe)
        self._split_probes = waiting[-3:]
        if lazy:
            self._adopt_split()
        cur = self._critic_split_bal if bal else self._critic_split
        n1 = min(E, cur) if cur else E
        ev[0].record(main)
        cs = main if cstream is None else cstream
        wg = self.critic_workgroups if cstream is None \
            else 8 * self.critic_cus_per_xcd
        if cstream is not None:
            cstream.wait_event(ev[0])
        # (ev[6]: behind the first few epochs -- a per-epoch time that is complete
        # long before the host comes back for the next split, lazy steps)
        nprobe = min(n1, 6)
        ev.append(torch.cuda.Event(enable_timing=True))
        with torch.cuda.stream(cs):
            ce.run(nprobe, wg)
            ev[6].record(cs)
            ce.run(n1 - nprobe, wg)
            ev[4].record(cs)
        side.wait_event(ev[0])
        with torch.cuda.stream(side):
            ev[2].record(side)
            policy_loss_dict = self._update_policy_beside_critic(dataset)
            ev[3].record(side)
            finish_side = side_work() if side_work is not None else dict
            ev[5].record(side)
        with torch.cuda.stream(cs):
            if n1 < E:
                cs.wait_event(ev[5])           # the policy stream is drained
                ce.run(E - n1, 0 if cstream is None else wg)
            ev[1].record(cs)
        main.wait_stream(side)
        if cstream is not None:
            main.wait_stream(cstream)
        if lazy:
            # nothing is read here: the caller's metrics resolve `tail` later;
            # the next split comes from this step's events once they are done
            # (looked at when the next update starts)
            self._split_probes = self._split_probes[-2:] + [(ev, n1, E, bal)]
            # every epoch is enqueued: the closure below must not keep the
            # rollout buffer alive (x / returns / old_values are views of it;
            # an unread LazyMetrics would pin ~0.25 GB per step at C2)
            ce.x = ce.returns = ce.old_values = None

            def tail():
                return ce.finish(), policy_loss_dict, \
                    ev[0].elapsed_time(ev[1]) * 1e-3, \
                    ev[2].elapsed_time(ev[3]) * 1e-3, finish_side()
            return tail
        critic_loss_dict = ce.finish()
        side_result = finish_side()
        torch.cuda.synchronize()
        # next split: the critic epochs the policy stream needs company for
        first_ms = ev[0].elapsed_time(ev[4]) / max(n1, 1)
        side_ms = ev[2].elapsed_time(ev[5])
        if self.adaptive_critic_split and cstream is None:
            split = int(min(E, side_ms / first_ms + self._split_margin))
            if self.dist.active:
                # every rank must issue its collectives in the same order (the
                # critic's first part, the policy's, the critic's rest): agree
                # on the largest split
                import torch.distributed as dist
                from ..dist import all_reduce
                t = torch.tensor([split], device=self.device)
                all_reduce(t, op=dist.ReduceOp.MAX, group=self.dist.group)
                split = int(t.item())
            if bal:
                self._critic_split_bal = split
            else:
                self._critic_split = split
        return critic_loss_dict, policy_loss_dict, \
            ev[0].elapsed_time(ev[1]) * 1e-3, \
            ev[2].elapsed_time(ev[3]) * 1e-3, side_result

    def _adopt_split(self):
        """Lazy steps: this rank's own estimate of the split (from its events)
        becomes the split -- directly in one process; with the envs sharded over
        ranks every rank must issue its collectives in the same order (the
        critic's first part, the policy's, the critic's rest), so the ranks
        agree on the LARGEST estimate without the host waiting for anything:
        each lazy step puts one MAX all-reduce of the two estimates (ordinary /
        balance-check iterations) on the main stream, followed by a copy into
        pinned host memory, and adopts the result of the exchange issued TWO
        steps earlier -- that step has been retired (its end event waited for),
        so the values are there, and every rank adopts the same exchange at the
        same step.  (Round 3: a blocking MAX all-reduce + .item() at the end of
        every step, and no lazy step at all in sharded runs.)"""
        if not self.dist.active:
            if self._local_split[0]:
                self._critic_split = self._local_split[0]
            if self._local_split[1]:
                self._critic_split_bal = self._local_split[1]
            return
        import torch.distributed as dist
        from ..dist import all_reduce
        if self._early_split_exchange():
            # The first iterations of a sharded run exchange the estimate at once
            # (the caller has waited for the device, so the previous step's
            # events have given it): the pipelined exchange below hands the
            # first measured split to iteration 6 -- until then every critic
            # epoch would run on 224 workgroups (+ 10 % per step), and a short
            # warm-up would time exactly those steps.
            t = torch.tensor(self._local_split, dtype=torch.int32,
                             device=self.device)
            all_reduce(t, op=dist.ReduceOp.MAX, group=self.dist.group)
            a, b = (int(v) for v in t.tolist())
            if a:
                self._critic_split = a
            if b:
                self._critic_split_bal = b
            return
        q = self._split_exchanges
        if len(q) >= 2:
            ev, host = q.pop(0)
            ev.synchronize()                    # long done (two steps ago)
            a, b = int(host[0]), int(host[1])
            if a:
                self._critic_split = a
            if b:
                self._critic_split_bal = b
        t = torch.tensor(self._local_split, dtype=torch.int32).pin_memory() \
            .to(self.device, non_blocking=True)
        all_reduce(t, op=dist.ReduceOp.MAX, group=self.dist.group)
        host = torch.empty(2, dtype=torch.int32).pin_memory()
        host.copy_(t, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        q.append((ev, host))

    def _early_split_exchange(self):
        """Sharded lazy steps 2 .. 6: the split is agreed with a blocking
        exchange (same decision on every rank: the iteration count)."""
        return self.dist.active and self.adaptive_critic_split and \
            self.num_iterations <= 6

    def _balance_iteration(self):
        """Is the current iteration one with the policy balance check
        (temporal_correlated_agent.py:447-451)?"""
        return isinstance(self.balance_check, int) and \
            not isinstance(self.balance_check, bool) and \
            self.num_iterations % self.balance_check == 1

    def _objective_streams(self):
        """The fused objective's second stream only where hardware queues are
        to spare: a sharded run already drives the critic stream, the policy
        stream and the streams of two RCCL communicators, and streams that
        share a hardware queue wait for each other's kernels (measured: the
        K x K kernels queued behind 2 ms critic launches, 2.7 ms per epoch)."""
        from .._lib import call
        n = int(os.environ.get("TCE_OBJECTIVE_STREAMS", "0")) or \
            (1 if self.dist.active and self.xchg_policy is None else 2)
        if n == 2 and self.device.type == "cuda":
            # a stream that is PROBED to run beside the main and the policy
            # stream (a stream that merely exists may share their hardware queue)
            from .. import streams
            streams.objective_side(self.device)
        call("tce_policy_objective_streams", n)

    def _update_policy_beside_critic(self, dataset):
        """The policy update while the critic's persistent grid holds most of
        the chip: tell the library so (tce_set_cu_budget), its kernels then
        prefer few full waves over many short ones."""
        from .._lib import call
        call("tce_set_cu_budget", max(256 - self.critic_workgroups, 16))
        try:
            return self.update_policy(dataset)
        finally:
            call("tce_set_cu_budget", 0)

    def _make_update_streams(self):
        """Second stream for the policy epochs; with ``critic_cus_per_xcd`` both
        updates get streams bound to disjoint compute units (the critic the
        units [32 - n, 32) of every XCD, the policy the rest)."""
        n = self.critic_cus_per_xcd
        if not n:
            from .. import streams
            self._policy_stream, self._critic_stream = \
                streams.get("policy", self.device), None
            return
        import ctypes
        from .. import _lib
        lib = _lib.load()
        hs = []
        for first, cnt in ((32 - n, n), (0, 32 - n)):
            h = ctypes.c_void_p()
            if lib.tce_stream_create_cu_range(first, cnt, ctypes.byref(h)):
                raise RuntimeError(lib.tce_last_error().decode())
            hs.append(h.value)
        self._critic_stream = torch.cuda.ExternalStream(hs[0])
        self._policy_stream = torch.cuda.ExternalStream(hs[1])

    # ---- dataset processing (GAE + segment advantage: HIP kernels) -----------
    def process_dataset(self, dataset):
        rewards, values = dataset["step_rewards"], dataset["step_values"]
        pred_pairs = self.sampler.pred_pairs
        fuse = self.segment_advantage == "value_subtraction"
        res = ops.gae(rewards, values, dataset["step_dones"],
                      dataset["step_time_limit_dones"], self.discount_factor,
                      self.gae_scaling, self.use_gae,
                      pred_pairs if fuse else None)
        dataset["step_advantages"], dataset["step_returns"] = res[0], res[1]
        dataset["segment_advantage"] = self.get_segment_advantage(
            rewards, values, res[0], pred_pairs,
            fused=(res[2], res[3]) if fuse else None)
        return dataset

    def get_advantage_return(self, rewards, values, dones, time_limit_dones):
        return ops.gae(rewards, values, dones, time_limit_dones,
                       self.discount_factor, self.gae_scaling, self.use_gae)

    def get_segment_advantage(self, rewards, values, advantages, pred_pairs,
                              fused=None, **kwargs):
        return ops.segment_advantage(
            self.segment_advantage, rewards, values, advantages, pred_pairs,
            self.discount_factor, self.norm_advantages,
            float(self.clip_advantages or 0.0), group=self.dist.group,
            fused=fused)

    # ---- critic ----------------------------------------------------------------
    def update_critic(self, dataset, defer=False, max_workgroups=0):
        D2 = self.policy.num_dof * 2
        states = dataset["step_states"]                  # [N, T, D] (view)
        N, T = states.shape[:2]
        old_values = dataset["step_values"][:, :-1]
        returns = dataset["step_returns"]
        from .. import critic_ops
        fused = critic_ops.supported(self.critic.net) and \
            self._critic_minibatches_fused()
        if fused:
            finish = self._update_critic_fused(states[..., :-D2], returns,
                                               old_values, max_workgroups)
            return finish if defer else finish()
        losses, norms, norms_c = [], [], []
        for _ in range(self.epochs_critic):
            for sel in self._minibatches(N * T):
                if sel is None:
                    s_in = states[..., :-D2]
                    v_old, ret = old_values, returns
                else:
                    s_in = states.reshape(N * T, -1)[sel][..., :-D2]
                    v_old = old_values.reshape(-1)[sel]
                    ret = returns.reshape(-1)[sel]
                values_new = self.critic.critic(s_in).squeeze(-1)
                loss = self.value_loss(values_new, ret, v_old)
                self.critic_optimizer.zero_grad(set_to_none=True)
                loss.backward()
                g, gc = self._optimizer_step(self.critic_optimizer,
                                             self.critic_net_params,
                                             self.clip_grad_norm)
                losses.append(loss.detach())
                norms.append(g)
                norms_c.append(gc)
        stacked = torch.stack([torch.stack(losses), torch.stack(norms),
                               torch.stack(norms_c)])

        def finish():
            host = stacked.cpu().numpy()
            return {**util.generate_stats(host[0], "critic_loss"),
                    **util.generate_stats(host[1], "critic_grad_norm"),
                    **util.generate_stats(host[2],
                                          "clipped_critic_grad_norm")}
        return finish if defer else finish()

    def _update_critic_fused(self, x, returns, old_values, max_workgroups=0):
        ce = _CriticEpochs(self, x, returns, old_values)
        ce.run(self.epochs_critic, max_workgroups)
        return ce.finish

    def _minibatches(self, n):
        """generate_minibatches (util_data_structure.py:378-391).  With ONE
        minibatch the permutation does not change the full-batch mean loss, so
        no gather is done (and the numpy generator is not consumed)."""
        if self.num_minibatchs == 1:
            return [None]
        idx = np.arange(n)
        np.random.shuffle(idx)
        return [torch.as_tensor(s, device=self.device)
                for s in np.array_split(idx, self.num_minibatchs)]

    # ---- policy ------------------------------------------------------------------
    def update_policy(self, dataset):
        self._objective_streams()
        D2 = self.policy.num_dof * 2
        states = dataset["segment_state"][..., :-D2]
        actions = dataset["step_actions"]
        log_probs_old = dataset["segment_log_prob_estimate"]
        mean_old = dataset["segment_params_mean"]
        L_old = dataset["segment_params_L"]
        seg_adv = dataset["segment_advantage"]
        init_time = dataset["segment_init_time"]
        init_pos = dataset["segment_init_pos"]
        init_vel = dataset["segment_init_vel"]
        times = self.sampler.get_times(init_time, self.sampler.num_times)
        pred_pairs = self.sampler.pred_pairs

        if self.projection.initial_entropy is None:
            ent0 = self.policy.entropy([mean_old, L_old]).mean()
            self.projection.initial_entropy = self.dist.mean_scalar(ent0)

        self.check_policy_balance = self._balance_iteration()

        def forward():
            mean_new, L_new = self.policy.policy(states)
            proj = self.projection(self.policy, (mean_new, L_new),
                                   (mean_old, L_old), self.num_iterations)
            return mean_new, L_new, proj[0], proj[1]

        def lp(proj_mean, proj_L):
            return self.policy.log_prob(
                actions, params_mean=proj_mean, params_L=proj_L, times=times,
                init_time=init_time, init_pos=init_pos, init_vel=init_vel,
                pred_pairs=pred_pairs)

        # per-epoch record: 7 loss/norm scalars, 12 KL terms, 3 NaN flags, the
        # two gradient norms of a balance-check epoch
        E = self.epochs_policy
        rec_all = torch.zeros(E, 24, dtype=self.dtype, device=self.device)
        rec_idx = torch.zeros(1, dtype=torch.int64, device=self.device)
        surr_gn, tr_gn = [], []

        # the fused objective (one C call) where it applies; the epochs of a
        # balance-check iteration need the whole epoch in C (DirectEpoch splits
        # the objective's gradient), else they run op by op
        use_fused = self.fused_policy_objective and \
            objective.supported(self, dataset)
        use_direct = use_fused and self.direct_policy_epoch and \
            not self.graph_policy_update and \
            objective.DirectEpoch.supported(self, states)
        if self.check_policy_balance and not (
                use_direct and (not self.dist.active
                                or self.xchg_policy is not None)):
            use_fused = use_direct = False
        fused_ctx = None
        if use_fused:
            init = self.projection.initial_entropy
            sched = self.projection.entropy_schedule_type
            beta = None if sched in (None, False) else \
                self.projection.entropy_schedule(
                    init, self.projection.target_entropy,
                    self.projection.temperature, self.num_iterations)
            fused_ctx = objective.Context(self, dataset, times, beta)

        direct = None
        if use_direct:
            direct = objective.DirectEpoch(self, states, fused_ctx)
        epoch_no = [0]
        balance_direct = direct is not None and self.check_policy_balance

        def epoch_fused():
            if direct is not None:
                # no autograd, no device-side record index: the epoch number
                # is known on the host (NaN flags are derived on the host too)
                row = rec_all[epoch_no[0]]
                direct.run(row[:19], balance=balance_direct, bal=row[22:24])
                epoch_no[0] += 1
                return
            mean_new, L_new = self.policy.policy(states)
            policy_loss, rec17 = objective.policy_objective(mean_new, L_new,
                                                            fused_ctx)
            self.policy_optimizer.zero_grad(set_to_none=True)
            policy_loss.backward()
            g, gc = self._optimizer_step(self.policy_optimizer,
                                         self.policy_net_params,
                                         self.clip_grad_norm)
            rec = torch.cat([rec17[:5], torch.stack([g, gc]).to(rec17.dtype),
                             rec17[5:], torch.isnan(rec17[:3]).to(rec17.dtype)])
            rec_all[:, :22].index_copy_(0, rec_idx, rec[None])
            rec_idx.add_(1)

        def epoch():
            if fused_ctx is not None:
                return epoch_fused()
            if self.check_policy_balance:
                mean_new, L_new, pm, pL = forward()
                s_loss, _ = self.surrogate_loss(seg_adv, lp(pm, pL),
                                                log_probs_old)
                self.policy_optimizer.zero_grad(set_to_none=True)
                s_loss.backward()
                surr_gn.append(self._grad_norm_clip(
                    0.0, self.policy_net_params)[0])
                mean_new, L_new, pm, pL = forward()
                t_loss = self.projection.get_trust_region_loss(
                    self.policy, (mean_new, L_new), (pm, pL),
                    set_variance=self.set_variance)
                self.policy_optimizer.zero_grad(set_to_none=True)
                t_loss.backward()
                tr_gn.append(self._grad_norm_clip(
                    0.0, self.policy_net_params)[0])

            mean_new, L_new, proj_mean, proj_L = forward()
            log_prob_new = lp(proj_mean, proj_L)
            surrogate_loss, ratio = self.surrogate_loss(
                seg_adv, log_prob_new, log_probs_old)
            with torch.no_grad():
                kl_row = self.kl_old_new_proj(
                    mean_new, L_new, mean_old, L_old, proj_mean, proj_L)
            entropy = self.policy.entropy([proj_mean, proj_L]).mean()
            entropy_loss = -self.entropy_penalty_coef * entropy
            trust_region_loss = self.projection.get_trust_region_loss(
                self.policy, (mean_new, L_new), (proj_mean, proj_L),
                set_variance=self.set_variance)
            policy_loss = surrogate_loss + entropy_loss + trust_region_loss
            self.policy_optimizer.zero_grad(set_to_none=True)
            policy_loss.backward()
            g, gc = self._optimizer_step(self.policy_optimizer,
                                         self.policy_net_params,
                                         self.clip_grad_norm)
            losses = torch.stack([surrogate_loss.detach(),
                                  entropy_loss.detach(),
                                  trust_region_loss.detach()])
            rec = torch.cat([losses, torch.stack([policy_loss.detach(),
                                                  entropy.detach(), g, gc]),
                             kl_row.to(losses.dtype),
                             torch.isnan(losses).to(losses.dtype)])
            rec_all[:, :22].index_copy_(0, rec_idx, rec[None])
            rec_idx.add_(1)

        util.run_time_test(lock=True, key="projection", sync=False)
        ev_a, ev_b = torch.cuda.Event(enable_timing=True), \
            torch.cuda.Event(enable_timing=True)
        ev_a.record()
        if self.graph_policy_update and not self.dist.active and E > 2 \
                and not self.check_policy_balance:
            # The epochs are identical launch sequences on fixed buffers: run
            # the first one eagerly, record the second into a HIP graph and
            # replay it -- ~150 launches per epoch leave the host.
            epoch()
            graph = self._capture(epoch)
            for _ in range(E - 1):
                graph.replay()
            self.policy_optimizer.host_step += E - 2   # capture counted one
            self._last_policy_graph = graph       # alive until the replays ran
        else:
            for _ in range(E):
                epoch()
        ev_b.record()
        projection_time = util.run_time_test(lock=False, key="projection",
                                             sync=False)

        if self.dist.active:
            # (every rank raises together: the flags of all shards in row 0)
            rec_all[0, 19:22] = self._nan_over_ranks(rec_all[:, :3]) \
                .to(rec_all.dtype)
        rec_host = rec_all.cpu().numpy()                  # ONE copy
        if direct is not None and not self.dist.active:
            rec_host[:, 19:22] = np.isnan(rec_host[:, :3])
        self._raise_on_nan(rec_host[:, 19:22].any(axis=0))
        host, kl_host = rec_host[:, :7], rec_host[:, 7:19]
        names = ("surrogate_loss", "entropy_loss", "trust_region_loss",
                 "policy_loss", "entropy", "policy_grad_norm",
                 "clipped_policy_grad_norm")
        out = {}
        for i, n in enumerate(names):
            out.update(util.generate_stats(host[:, i], n))
        kl_names = [a + "_" + b for a in ("new_old", "new_proj", "proj_old")
                    for b in ("mean_diff", "cov_diff", "shape_diff",
                              "volume_diff")]
        for i, n in enumerate(kl_names):
            out.update(util.generate_stats(kl_host[:, i], "projection_" + n))
        out["projection_time"] = projection_time
        out["policy_epochs_device_time"] = ev_a.elapsed_time(ev_b) * 1e-3
        if self.check_policy_balance:
            if balance_direct:
                sg, tg = rec_host[:, 22], rec_host[:, 23]
            else:
                sg = torch.stack(surr_gn).cpu().numpy()
                tg = torch.stack(tr_gn).cpu().numpy()
            out.update(util.generate_stats(sg, "surrogate_grad_norm"))
            out.update(util.generate_stats(tg, "trust_region_grad_norm"))
            with np.errstate(divide="ignore", invalid="ignore"):
                out["balance_ratio"] = float(
                    np.float64(out["surrogate_grad_norm_mean"]) /
                    np.float64(out["trust_region_grad_norm_mean"]))

        if self.set_variance and not self.policy.contextual_cov:
            with torch.no_grad():
                _, _, _, pL = forward()
                self.policy.set_cov_variable(pL)
        return out

    def kl_old_new_proj(self, mean_new, L_new, mean_old, L_old, proj_mean,
                        proj_L):
        """12 scalars (means over the batch) as one device vector."""
        parts = []
        for p, q in (((mean_new, L_new), (mean_old, L_old)),
                     ((mean_new, L_new), (proj_mean, proj_L)),
                     ((proj_mean, proj_L), (mean_old, L_old))):
            parts.extend(d.mean() for d in
                         gaussian_kl_details(self.policy, p, q))
        return torch.stack(parts)

    def value_loss(self, values, returns, old_vs):
        vf_loss = (returns - values).pow(2)
        if self.clip_critic > 0:
            vs_clipped = old_vs + (values - old_vs).clamp(-self.clip_critic,
                                                          self.clip_critic)
            vf_loss = torch.max(vf_loss, (vs_clipped - returns).pow(2))
        return vf_loss.mean()

    @staticmethod
    def surrogate_loss(advantages, log_prob_new, log_prob_old):
        ratio = (log_prob_new - log_prob_old).exp()
        return -(ratio * advantages).mean(), ratio.mean().detach()

    def entropy_loss(self, params_mean, params_L):
        entropy = self.policy.entropy([params_mean, params_L]).mean()
        return -self.entropy_penalty_coef * entropy, {"entropy": entropy}

    def save_agent(self, log_dir, epoch):
        super().save_agent(log_dir, epoch)
        self.sampler.save_rms(log_dir, epoch)

    def load_agent(self, log_dir, epoch):
        super().load_agent(log_dir, epoch)
        self.sampler.load_rms(log_dir, epoch)


class _EpochGraph:
    """The E epochs of one update of the black-box agent as ONE HIP graph that
    is kept across iterations: the update's inputs live in static buffers that
    every iteration overwrites, the per-epoch record and its row counter are
    static too.  Recording the epoch anew in every iteration (and destroying
    the previous graph) cost 10 - 15 ms of host time per step -- these updates
    are host-bound."""

    def __init__(self, sig, inputs, rec_cols, E, dtype, device):
        self.sig, self.graph, self.static, self.last = sig, None, {}, {}
        for k, v in inputs.items():
            base = getattr(v, "_tce_base", None)
            if base is not None:                  # one factor shared by all envs
                buf = torch.empty_like(base)
                self.static[k] = (ops.expand_shared(buf, v.shape[0]), buf)
            else:
                buf = torch.empty_like(v, memory_format=torch.contiguous_format)
                self.static[k] = (buf, buf)
        self.rec = torch.zeros(E, rec_cols, dtype=dtype, device=device)
        self.idx = torch.zeros(1, dtype=torch.int64, device=device)

    def bind(self, inputs):
        """Copy this iteration's inputs into the static buffers."""
        out = {}
        for k, v in inputs.items():
            view, buf = self.static[k]
            base = getattr(v, "_tce_base", None)
            buf.copy_(base if base is not None else v)
            out[k] = view
        self.idx.zero_()
        return out


class BlackBoxAgent(TemporalCorrelatedAgent):
    """black_box_agent.py: episode-level advantage R - V(s0), critic regresses
    the episode return, param-space log-prob; otherwise the same update."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        # the epochs of both updates are ~100 launch-bound kernels each and
        # nothing else runs beside them: replay them from HIP graphs
        self.graph_epochs = kwargs.get("graph_epochs", True)
        # ... and keep the graphs across iterations (inputs in static buffers)
        self.cache_epoch_graphs = kwargs.get("cache_epoch_graphs", True)
        # the objective as ONE autograd node (tce_bb_policy_objective_*) is
        # an option here: with 32-wide nets and K = 20 the op-by-op graph is
        # host-bound at 0.27 ms per epoch, the fused one device-bound at 0.45
        # (its K x K kernels are single workgroups): 24 vs 27 ms per step
        self.fused_policy_objective = kwargs.get("fused_policy_objective",
                                                 False)
        # the hand-written row kernels for nets up to 64 wide (csrc/smlp.hip):
        # one launch per critic epoch, six per policy epoch, no autograd, no
        # library GEMM, no graph.  Off: the op-by-op / graph paths below.
        self.small_net_kernels = kwargs.get("small_net_kernels", True)
        self.lazy_metrics = kwargs.get(
            "lazy_metrics", os.environ.get("TCE_LAZY_METRICS", "1") != "0")
        self._epoch_graphs = {}

    def _epoch_graph(self, kind, E, opt, inputs, rec_cols):
        """-> (_EpochGraph or None, inputs to use).  None: the update records
        its epochs anew (or launches them eagerly) as before.  A graph is kept
        while nothing the recording baked in changes: shapes, learning rate,
        and -- a projection with an entropy schedule computes its bound from
        the iteration number on the host -- only without such a schedule."""
        if not (self.graph_epochs and self.cache_epoch_graphs and E > 2 and
                self.num_minibatchs == 1 and not self.dist.active and
                self.projection.entropy_schedule_type in (None, False)):
            return None, inputs
        g = opt.param_groups[0]
        pr = self.projection
        sig = (E, g["lr"], g.get("weight_decay", 0.0), tuple(g["betas"]),
               g["eps"], self.clip_grad_norm,
               self.clip_critic, self.entropy_penalty_coef, self.set_variance,
               # scalar kernel arguments of the projection the recording bakes in
               float(getattr(pr, "mean_bound", 0.0)),
               float(getattr(pr, "cov_bound", 0.0)),
               float(getattr(pr, "trust_region_coeff", 0.0)),
               tuple((k, tuple(v.shape), v.dtype,
                      getattr(v, "_tce_base", None) is not None)
                     for k, v in inputs.items()))
        eg = self._epoch_graphs.get(kind)
        if eg is None or eg.sig != sig:
            if eg is not None:
                n = self._graph_rerecords = getattr(
                    self, "_graph_rerecords", 0) + 1
                if n == 3:
                    import warnings
                    warnings.warn(
                        "BlackBoxAgent: the kept %s epoch graph was re-recorded "
                        "3 times (a learning-rate schedule or changing bounds "
                        "invalidate it every iteration): the saving of "
                        "cache_epoch_graphs is lost" % kind)
            eg = _EpochGraph(sig, inputs, rec_cols, E, self.dtype, self.device)
            self._epoch_graphs[kind] = eg
        return eg, eg.bind(inputs)

    def load_agent(self, log_dir, epoch):
        super().load_agent(log_dir, epoch)
        self._epoch_graphs = {}         # recorded against the old state

    def _run_epoch_graph(self, eg, epoch, E, opt):
        """Replay (or, the first time, record) the kept graph E times."""
        if eg.graph is None:
            n = E
            if not getattr(opt, "_tce_graph_warm", False):
                epoch()                           # see _run_epochs
                opt._tce_graph_warm = True
                n = E - 1
            eg.graph = self._capture(epoch, pool_key=id(opt))
            opt.host_step -= 1                    # the recording counted one
        else:
            n = E
        for _ in range(n):
            eg.graph.replay()
        opt.host_step += n

    def _step_lazy(self):
        """step() that leaves its host reads to the returned metrics
        (util.LazyMetrics; TemporalCorrelatedAgent.lazy_metrics): the row-kernel
        updates of both networks side by side, nothing waited for."""
        self.num_iterations += 1
        self._retire_lazy_steps(2)
        done = self._lazy_done
        main = torch.cuda.current_stream()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record(main)
        dataset, n_steps = self.sampler.run(training=True, policy=self.policy,
                                            critic=self.critic)
        self.num_global_steps += n_steps * self.dist.world
        ev[1].record(main)
        dataset = self.process_dataset(dataset)
        stats_fin = util.device_stats_async(
            {k: v for k, v in dataset.items()
             if k not in ("segment_params_L", "segment_state")}, "exploration")
        if getattr(self, "_bb_stream", None) is None:
            from .. import streams
            self._bb_stream = streams.get("policy", self.device)
        side = self._bb_stream
        side.wait_stream(main)
        # The policy's chain is the longer one (3 dependent kernels per epoch,
        # 38 - 40 us, against the critic's 2, 21 us): it is enqueued FIRST.  The
        # kernels of a chain run back to back once they are queued (the trace
        # shows no gap between them), so an update lasts as long as the policy
        # chain plus whatever the host spent before it reached its first
        # launch -- with the critic's 200 launches in front that was 1.1 ms of a
        # 5.7 ms update (scripts/rocpd_seq.py on a C4 trace).
        finish_policy = self.update_policy(dataset, defer=True)
        with torch.cuda.stream(side):
            finish_critic = self.update_critic(dataset, defer=True)
        main.wait_stream(side)
        ev[2].record(main)
        steps = self.num_global_steps

        def resolve():
            ev[2].synchronize()
            return {**stats_fin(), **finish_critic(), **finish_policy(),
                    "sampling_time": ev[0].elapsed_time(ev[1]) * 1e-3,
                    "update_time": ev[1].elapsed_time(ev[2]) * 1e-3,
                    "num_global_steps": steps, "lr_policy": self.lr_policy,
                    "lr_critic": self.lr_critic}
        result = util.LazyMetrics(resolve)
        done.append((ev[2], result))
        if self.evaluation_interval and (
                self.evaluation_interval == 1 or
                self.num_iterations % self.evaluation_interval == 1):
            evd = self.evaluate()[0]
            result.update(util.device_stats(
                {k: v for k, v in evd.items()
                 if k not in ("segment_params_L", "segment_state")},
                "evaluation"))
        return result

    def _critic_path(self):
        """Which hand-written critic update applies: "smlp" (nets up to 64
        wide, csrc/smlp.hip), "pmlp" (128 x 1 / 128 x 2 / 256 x 1 on the row
        kernels of csrc/pmlp.hip: table tennis's BBRL critic), "fused" (the
        matrix-core epochs of the TCE critics: box pushing's 256 x 2), None
        (op by op / HIP graph)."""
        from .. import critic_ops, pmlp_ops, smlp_ops
        if not self.small_net_kernels:
            return None
        if smlp_ops.critic_supported(self):
            return "smlp"
        if self.device.type != "cuda":
            return None
        net, opt = self.critic.net, self.critic_optimizer
        if critic_ops.supported(net) and smlp_ops._opt_matches(
                opt, list(net.parameters())) and \
                self._critic_minibatches_fused():
            return "fused"
        if pmlp_ops.critic_supported(self):
            return "pmlp"
        return None

    def _policy_path(self, dataset):
        """As _critic_path for the policy update: "smlp", "direct"
        (objective.BBDirectEpoch: the mean nets of csrc/pmlp.hip / the fused
        128 x 2 kernels) or None."""
        from .. import smlp_ops
        L_old = dataset["segment_params_L"]
        # (the policy update is full-batch whatever num_minibatchs says:
        # black_box_agent.py:159-389 has no generate_minibatches)
        if not self.small_net_kernels:
            return None
        if smlp_ops.policy_supported(self, L_old):
            return "smlp"
        if self.device.type == "cuda" and objective.BBDirectEpoch.supported(
                self, dataset["segment_state"], L_old):
            return "direct"
        return None

    def step(self):
        # (sharded runs too: the two updates issue their gradient all-reduces on
        # two communicators -- the critic's on the default group from the side
        # stream, the policy's on the agent's second group from the main stream --
        # in the same host order on every rank, and nothing waits for the device)
        if self.lazy_metrics and self.overlap_updates and \
                self.small_net_kernels and \
                self.device.type == "cuda" and \
                self.projection.initial_entropy is not None and \
                self._critic_path() is not None and \
                getattr(self, "_bb_small_policy", False):
            return self._step_lazy()
        self.num_iterations += 1
        util.run_time_test(lock=True, key="sampling")
        dataset, n_steps = self.sampler.run(training=True, policy=self.policy,
                                            critic=self.critic)
        self.num_global_steps += n_steps * self.dist.world
        sampling_time = util.run_time_test(lock=False, key="sampling")
        dataset = self.process_dataset(dataset)
        dataset_stats = util.device_stats(
            {k: v for k, v in dataset.items()
             if k not in ("segment_params_L", "segment_state")}, "exploration")
        util.run_time_test(lock=True, key="update")
        small = self._critic_path() is not None and \
            self._policy_path(dataset) is not None
        # (the lazy step needs both updates on the hand-written kernels -- no
        # graph, no autograd, deferrable reads: known from here on)
        self._bb_small_policy = bool(small)
        if self.overlap_updates and \
                (small or (self.num_minibatchs == 1 and self.graph_epochs
                           and not self.dist.active)):
            # the two updates are independent chains of ~100 small launches per
            # epoch, replayed from HIP graphs: side by side on two streams
            main = torch.cuda.current_stream()
            if getattr(self, "_bb_stream", None) is None:
                from .. import streams
                self._bb_stream = streams.get("policy", self.device)
            side = self._bb_stream
            side.wait_stream(main)
            if small:
                # (the longer chain first: see _step_lazy)
                finish_policy = self.update_policy(dataset, defer=True)
                with torch.cuda.stream(side):
                    finish_critic = self.update_critic(dataset, defer=True)
                main.wait_stream(side)
                policy_loss_dict = finish_policy()
            else:
                with torch.cuda.stream(side):
                    finish_critic = self.update_critic(dataset, defer=True)
                policy_loss_dict = self.update_policy(dataset)
                main.wait_stream(side)
            critic_loss_dict = finish_critic()
        else:
            critic_loss_dict = self.update_critic(dataset)
            policy_loss_dict = self.update_policy(dataset)
        update_time = util.run_time_test(lock=False, key="update")
        self.dist.check_exchanges()
        result = {**dataset_stats, **critic_loss_dict, **policy_loss_dict,
                  "sampling_time": sampling_time, "update_time": update_time,
                  "num_global_steps": self.num_global_steps,
                  "lr_policy": self.lr_policy, "lr_critic": self.lr_critic}
        if self.evaluation_interval and (
                self.evaluation_interval == 1 or
                self.num_iterations % self.evaluation_interval == 1):
            ev = self.evaluate()[0]
            result.update(util.device_stats(
                {k: v for k, v in ev.items()
                 if k not in ("segment_params_L", "segment_state")},
                "evaluation"))
        return result

    def process_dataset(self, dataset):
        adv = dataset["segment_reward"] - dataset["segment_value"]
        stats = ops.moments(adv, self.dist.group) \
            if self.norm_advantages else None
        if stats is not None or self.clip_advantages > 0:
            adv = ops.normalize(adv, stats, 1e-8,
                                float(self.clip_advantages or 0.0),
                                single_std_one=True)
        dataset["segment_advantage"] = adv
        return dataset

    def update_critic(self, dataset, defer=False):
        states = dataset["segment_state"]
        old_values, returns = dataset["segment_value"], \
            dataset["segment_reward"]
        E = self.epochs_critic
        stats = lambda host: {
            **util.generate_stats(host[0], "critic_loss"),
            **util.generate_stats(host[1], "critic_grad_norm"),
            **util.generate_stats(host[2], "clipped_critic_grad_norm")}
        from .. import pmlp_ops, smlp_ops
        path = self._critic_path()
        if path == "smlp":
            # E launches, each a whole epoch incl. the Adam step (csrc/smlp.hip)
            rec = smlp_ops.critic_update(self, states, returns, old_values)
            fin = lambda: stats(rec.cpu().numpy().T)
            return fin if defer else fin()
        if path == "pmlp":
            # one C call per epoch on the row kernels of csrc/pmlp.hip
            rec = pmlp_ops.critic_update(self, states, returns, old_values)
            fin = lambda: stats(rec.cpu().numpy().T)
            return fin if defer else fin()
        if path == "fused":
            # the matrix-core epochs of the TCE critics (rows = envs)
            ce = _CriticEpochs(self, states, returns, old_values)
            ce.run(E)
            return ce.finish if defer else ce.finish()
        if self.num_minibatchs == 1:
            eg, st = self._epoch_graph(
                "critic", E, self.critic_optimizer,
                dict(states=states, returns=returns, old_values=old_values), 3)
            states, returns, old_values = st["states"], st["returns"], \
                st["old_values"]
            # per-epoch record {loss, |g|, |g| clipped}, written on the device
            rec = eg.rec if eg else torch.zeros(E, 3, dtype=self.dtype,
                                                device=self.device)
            idx = eg.idx if eg else torch.zeros(1, dtype=torch.int64,
                                                device=self.device)

            def epoch():
                loss = self.value_loss(
                    self.critic.critic(states).squeeze(-1), returns,
                    old_values)
                self.critic_optimizer.zero_grad(set_to_none=True)
                loss.backward()
                g, gc = self._optimizer_step(self.critic_optimizer,
                                             self.critic_net_params,
                                             self.clip_grad_norm)
                rec.index_copy_(0, idx, torch.stack(
                    [loss.detach(), g, gc])[None])
                idx.add_(1)

            if eg:
                self._run_epoch_graph(eg, epoch, E, self.critic_optimizer)
            else:
                self._run_epochs(epoch, E, self.critic_optimizer,
                                 self.graph_epochs)
            if defer:             # the host read waits for the caller's join
                return lambda: stats(rec.cpu().numpy().T)
            host = rec.cpu().numpy().T
        else:
            losses, norms, norms_c = [], [], []
            for _ in range(E):
                for sel in self._minibatches(states.shape[0]):
                    s_in, v_old, ret = states[sel], old_values[sel], \
                        returns[sel]
                    loss = self.value_loss(
                        self.critic.critic(s_in).squeeze(-1), ret, v_old)
                    self.critic_optimizer.zero_grad(set_to_none=True)
                    loss.backward()
                    g, gc = self._optimizer_step(self.critic_optimizer,
                                                 self.critic_net_params,
                                                 self.clip_grad_norm)
                    losses.append(loss.detach())
                    norms.append(g)
                    norms_c.append(gc)
            host = torch.stack([torch.stack(losses), torch.stack(norms),
                                torch.stack(norms_c)]).cpu().numpy()
        return (lambda: stats(host)) if defer else stats(host)

    def update_policy(self, dataset, defer=False):
        states = dataset["segment_state"]
        actions = dataset["segment_action"]
        log_probs_old = dataset["segment_log_prob"]
        mean_old, L_old = dataset["segment_params_mean"], \
            dataset["segment_params_L"]
        seg_adv = dataset["segment_advantage"]
        if self.projection.initial_entropy is None:
            ent0 = self.policy.entropy([mean_old, L_old]).mean()
            self.projection.initial_entropy = self.dist.mean_scalar(ent0)
        E = self.epochs_policy
        path = self._policy_path(dataset)
        if path == "smlp":
            return self._update_policy_small(dataset, defer=defer)
        if path == "direct":
            return self._update_policy_direct(dataset, defer=defer)
        assert not defer, "deferred reads: hand-written epochs only"
        # per epoch: 7 loss / norm scalars + the 12 means of kl_old_new_proj
        eg, st = self._epoch_graph(
            "policy", E, self.policy_optimizer,
            dict(states=states, actions=actions, log_probs_old=log_probs_old,
                 mean_old=mean_old, L_old=L_old, seg_adv=seg_adv), 19)
        states, actions, log_probs_old = st["states"], st["actions"], \
            st["log_probs_old"]
        mean_old, L_old, seg_adv = st["mean_old"], st["L_old"], st["seg_adv"]
        rec = eg.rec if eg else torch.zeros(E, 19, dtype=self.dtype,
                                            device=self.device)
        idx = eg.idx if eg else torch.zeros(1, dtype=torch.int64,
                                            device=self.device)
        last = eg.last if eg else {}
        self._objective_streams()
        fused_ctx = None
        if self.fused_policy_objective and \
                objective.bb_supported(self, L_old):
            init = self.projection.initial_entropy
            sched = self.projection.entropy_schedule_type
            beta = None if sched in (None, False) else \
                self.projection.entropy_schedule(
                    init, self.projection.target_entropy,
                    self.projection.temperature, self.num_iterations)
            fused_ctx = last.get("ctx") if eg and eg.graph is not None else \
                None
            if fused_ctx is None:
                fused_ctx = objective.BBContext(self, mean_old, L_old, actions,
                                                log_probs_old, seg_adv, beta)
                last["ctx"] = fused_ctx

        def epoch_fused():
            # projection -> log-prob -> surrogate -> entropy / trust region
            # loss and their gradients as ONE autograd node (one C call)
            mean_new, L_new = self.policy.policy(states)
            policy_loss, rec17 = objective.policy_objective(mean_new, L_new,
                                                            fused_ctx)
            self.policy_optimizer.zero_grad(set_to_none=True)
            policy_loss.backward()
            g, gc = self._optimizer_step(self.policy_optimizer,
                                         self.policy_net_params,
                                         self.clip_grad_norm)
            rec.index_copy_(0, idx, torch.cat(
                [rec17[:5], torch.stack([g, gc]).to(rec17.dtype),
                 rec17[5:17]])[None])
            idx.add_(1)
            last["t"] = (mean_new.detach(), ops.detach_L(L_new),
                         fused_ctx.proj_mean,
                         ops.expand_shared(fused_ctx.proj_L, states.shape[0]))

        def epoch():
            if fused_ctx is not None:
                return epoch_fused()
            mean_new, L_new = self.policy.policy(states)
            proj_mean, proj_L = self.projection(
                self.policy, (mean_new, L_new), (mean_old, L_old),
                self.num_iterations)
            log_prob_new = self.policy.log_prob(actions, params_mean=proj_mean,
                                                params_L=proj_L)
            surrogate_loss, _ = self.surrogate_loss(seg_adv, log_prob_new,
                                                    log_probs_old)
            with torch.no_grad():           # black_box_agent.py:308-310
                kl_row = self.kl_old_new_proj(
                    mean_new, L_new, mean_old, L_old, proj_mean, proj_L)
            entropy = self.policy.entropy([proj_mean, proj_L]).mean()
            entropy_loss = -self.entropy_penalty_coef * entropy
            trust_region_loss = self.projection.get_trust_region_loss(
                self.policy, (mean_new, L_new), (proj_mean, proj_L),
                set_variance=self.set_variance)
            policy_loss = surrogate_loss + entropy_loss + trust_region_loss
            self.policy_optimizer.zero_grad(set_to_none=True)
            policy_loss.backward()
            g, gc = self._optimizer_step(self.policy_optimizer,
                                         self.policy_net_params,
                                         self.clip_grad_norm)
            rec.index_copy_(0, idx, torch.cat([torch.stack([
                surrogate_loss.detach(), entropy_loss.detach(),
                trust_region_loss.detach(), policy_loss.detach(),
                entropy.detach(), g, gc]), kl_row.to(self.dtype)])[None])
            idx.add_(1)
            # the last epoch's distributions (fixed graph buffers when replayed)
            last["t"] = (mean_new.detach(), ops.detach_L(L_new),
                         proj_mean.detach(), ops.detach_L(proj_L))

        if eg:
            self._run_epoch_graph(eg, epoch, E, self.policy_optimizer)
        else:
            self._run_epochs(epoch, E, self.policy_optimizer,
                             self.graph_epochs)
        return self._finish_policy_update(rec, last["t"], states, mean_old,
                                          L_old)

    def _update_policy_small(self, dataset, defer=False):
        """update_policy on the row kernels of csrc/smlp.hip: per epoch the
        Cholesky head, the covariance projection, ONE kernel for everything
        per env (mean net forward, mean projection, log-prob, surrogate, trust
        region, their gradients, mean net backward), the K x K KL parts, the
        projection's backward and a finish kernel (Cholesky head backward,
        clip, Adam, record row)."""
        from .. import smlp_ops
        states = dataset["segment_state"]
        mean_old, L_old = dataset["segment_params_mean"], \
            dataset["segment_params_L"]
        sched = self.projection.entropy_schedule_type
        beta = None if sched in (None, False) else \
            self.projection.entropy_schedule(
                self.projection.initial_entropy,
                self.projection.target_entropy, self.projection.temperature,
                self.num_iterations)
        if beta is not None and not torch.is_tensor(beta):
            beta = torch.as_tensor(float(beta), device=self.device)
        # (env shards: the two norms are those of the rank-averaged parts)
        balance = self._balance_iteration()
        self.check_policy_balance = balance
        rec, mean_new, L_new, proj_mean, proj_L = smlp_ops.policy_update(
            self, states, dataset["segment_action"],
            dataset["segment_log_prob"], dataset["segment_advantage"],
            mean_old, L_old, beta, balance=balance)
        N = states.shape[0]
        last = (mean_new, ops.expand_shared(L_new, N), proj_mean,
                ops.expand_shared(proj_L, N))
        return self._finish_policy_update(rec, last, states, mean_old, L_old,
                                          defer=defer, balance=balance)

    def _update_policy_direct(self, dataset, defer=False):
        """update_policy for the mean nets of csrc/pmlp.hip / the fused 128 x 2
        kernels: every epoch ONE C call (objective.BBDirectEpoch), the epochs
        of a balance-check iteration (black_box_agent.py:218-284) included."""
        states = dataset["segment_state"]
        mean_old, L_old = dataset["segment_params_mean"], \
            dataset["segment_params_L"]
        sched = self.projection.entropy_schedule_type
        beta = None if sched in (None, False) else \
            self.projection.entropy_schedule(
                self.projection.initial_entropy,
                self.projection.target_entropy, self.projection.temperature,
                self.num_iterations)
        if beta is not None and not torch.is_tensor(beta):
            beta = torch.as_tensor(float(beta), device=self.device)
        self._objective_streams()
        ctx = objective.BBContext(self, mean_old, L_old,
                                  dataset["segment_action"],
                                  dataset["segment_log_prob"],
                                  dataset["segment_advantage"], beta)
        direct = objective.BBDirectEpoch(self, states, ctx)
        # (env shards without the in-library exchange stop the call in front of
        # the step and cannot split the epoch: their balance norms are left out)
        balance = self._balance_iteration() and (
            not self.dist.active or self.xchg_policy is not None)
        self.check_policy_balance = balance
        E, N = self.epochs_policy, states.shape[0]
        # per epoch: 7 loss / norm scalars, 12 KL means, the two balance norms
        rec = torch.zeros(E, 21, dtype=self.dtype, device=self.device)
        for e in range(E):
            direct.run(rec[e, :19], balance=balance, bal=rec[e, 19:21],
                       last=e == E - 1)
        mean_new, L_new = direct.latest()
        last = (mean_new, ops.expand_shared(L_new, N), ctx.proj_mean,
                ops.expand_shared(ctx.proj_L, N))
        return self._finish_policy_update(rec, last, states, mean_old, L_old,
                                          defer=defer, balance=balance)

    def _finish_policy_update(self, rec, last, states, mean_old, L_old,
                              defer=False, balance=False):
        """Everything that changes device state is enqueued here; the host
        reads (per-epoch record, projection metrics) happen in the returned
        closure when `defer` (BlackBoxAgent's lazy step), else at once."""
        mean_new, L_new, proj_mean, proj_L = last
        metrics = self.projection.compute_metrics(
            self.policy, (mean_new, L_new), (proj_mean, proj_L),
            self.num_iterations)
        mkeys = list(metrics.keys())
        mdev = torch.stack([v.to(self.dtype) for v in metrics.values()])
        if self.set_variance and not self.policy.contextual_cov:
            with torch.no_grad():
                m, L = self.policy.policy(states)
                _, pL = self.projection(self.policy, (m, L),
                                        (mean_old, L_old),
                                        self.num_iterations)
                self.policy.set_cov_variable(pL)

        # (env shards: the flags of every rank, enqueued here in step with the
        # peers -- the deferred host read below must not issue a collective)
        gflags = self._nan_over_ranks(rec[:, :3]) if self.dist.active else None

        def read():
            host = rec.cpu().numpy()                      # ONE copy
            bad = np.isnan(host[:, :3]).any(axis=0)
            if gflags is not None:
                bad = bad | (gflags.cpu().numpy() > 0)
            self._raise_on_nan(bad)
            names = ("surrogate_loss", "entropy_loss", "trust_region_loss",
                     "policy_loss", "entropy", "policy_grad_norm",
                     "clipped_policy_grad_norm")
            out = {}
            for i, n in enumerate(names):
                out.update(util.generate_stats(host[:, i], n))
            mh = mdev.cpu().numpy()
            out.update({"projection_" + k: float(v)
                        for k, v in zip(mkeys, mh)})
            if host.shape[1] >= 19:
                # kl_old_new_proj (black_box_agent.py:391-436) per epoch
                kl_names = [a + "_" + b
                            for a in ("new_old", "new_proj", "proj_old")
                            for b in ("mean_diff", "cov_diff", "shape_diff",
                                      "volume_diff")]
                for i, n in enumerate(kl_names):
                    out.update(util.generate_stats(host[:, 7 + i],
                                                   "projection_" + n))
            if balance:
                out.update(util.generate_stats(host[:, 19],
                                               "surrogate_grad_norm"))
                out.update(util.generate_stats(host[:, 20],
                                               "trust_region_grad_norm"))
                with np.errstate(divide="ignore", invalid="ignore"):
                    out["balance_ratio"] = float(
                        np.float64(out["surrogate_grad_norm_mean"]) /
                        np.float64(out["trust_region_grad_norm_mean"]))
            return out
        return read if defer else read()


def agent_factory(typ, **kwargs):
    return {"TemporalCorrelatedAgent": TemporalCorrelatedAgent,
            "BlackBoxAgent": BlackBoxAgent}[typ](**kwargs)
