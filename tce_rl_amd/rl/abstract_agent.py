"""AbstractAgent: mirror of mprl/rl/agent/abstract_agent.py:12-255 -- optimizers,
schedulers, checkpoints, evaluation, the env-shard gradient exchange -- shared
by the two agents (rl/tce_agent.py, rl/bb_agent.py).  See rl/agent.py for the
overview."""
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
