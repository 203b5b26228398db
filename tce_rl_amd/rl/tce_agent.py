"""TemporalCorrelatedAgent: mirror of
mprl/rl/agent/temporal_correlated_agent.py:12-753.  See rl/agent.py for the
overview."""
import collections
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
from .abstract_agent import AbstractAgent
from .critic_epochs import CriticEpochs as _CriticEpochs


# What an update will run on, decided in ONE place per update and returned as a
# named plan (the update itself, the tests' path assertions and bench.py read
# it; ``last_critic_plan`` / ``last_policy_plan`` keep the latest).
#   CriticPlan.kind: "fused-narrow" (csrc/mlp.hip, 128 x 2 float32),
#     "fused-wide" (csrc/mlpw_*.hip: 256 x 2, float64) or "autograd" (layer by
#     layer on csrc/glin.hip under torch autograd); minibatches: optimizer
#     steps per epoch; overlap: beside the policy update on a second stream;
#     lazy: no host wait at the end of step().
#   PolicyPlan.kind: "direct" (the epoch as ONE C call, no autograd), "node"
#     (the objective as one autograd node), "op_by_op"; balance: this
#     iteration carries the policy balance check; graph: epochs replayed from
#     a HIP graph.
CriticPlan = collections.namedtuple("CriticPlan",
                                    "kind minibatches overlap lazy")
PolicyPlan = collections.namedtuple("PolicyPlan", "kind balance graph")


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
        # sequential host shuffle), "device" = a keyed Feistel permutation
        # computed on the GPU (tce_feistel_permutation; not the reference's
        # sequence)
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

    def critic_plan(self):
        """THE path selection of update_critic (see CriticPlan)."""
        from .. import critic_ops
        net = self.critic.net
        k = int(self.num_minibatchs or 1)
        kind = "autograd"
        if self.device.type == "cuda" and critic_ops.supported(net) and \
                self._critic_minibatches_fused():
            kind = "fused-wide" if critic_ops.wide_supported(net) \
                else "fused-narrow"
        # side by side with the policy update: both must be enqueued without a
        # host read in between -- one minibatch, or minibatches inside the fused
        # epochs (the policy update is full-batch always:
        # temporal_correlated_agent.py:381-639)
        overlap = bool(self.overlap_updates) and (k == 1 or kind != "autograd")
        lazy = bool(self.lazy_metrics) and overlap and kind != "autograd"
        plan = CriticPlan(kind, k, overlap, lazy)
        self.last_critic_plan = plan
        return plan

    def policy_plan(self, dataset, states):
        """THE path selection of update_policy (see PolicyPlan)."""
        balance = self._balance_iteration()
        fused = self.fused_policy_objective and \
            objective.supported(self, dataset)
        direct = fused and self.direct_policy_epoch and \
            not self.graph_policy_update and \
            objective.DirectEpoch.supported(self, states)
        # the epochs of a balance-check iteration need the whole epoch in C
        # (DirectEpoch splits the objective's gradient), else they run op by op
        if balance and not (direct and (not self.dist.active
                                        or self.xchg_policy is not None)):
            fused = direct = False
        graph = bool(self.graph_policy_update) and not self.dist.active \
            and self.epochs_policy > 2 and not balance
        plan = PolicyPlan("direct" if direct else "node" if fused
                          else "op_by_op", balance, graph)
        self.last_policy_plan = plan
        return plan

    def _lazy_step_possible(self):
        return self.critic_plan().lazy

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
        return self.critic_plan().overlap

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
        fused = self.critic_plan().kind != "autograd"
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

        plan = self.policy_plan(dataset, states)
        use_direct = plan.kind == "direct"
        use_fused = plan.kind in ("direct", "node")
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
        if plan.graph:
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
