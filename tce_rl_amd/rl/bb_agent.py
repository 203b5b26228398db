"""BlackBoxAgent: mirror of mprl/rl/agent/black_box_agent.py:12-495.  See
rl/agent.py for the overview."""
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
from .critic_epochs import CriticEpochs as _CriticEpochs
from .tce_agent import CriticPlan, PolicyPlan, TemporalCorrelatedAgent


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

    def critic_plan(self):
        """The black-box agent's critic path as a named plan (kind: "smlp" /
        "pmlp" / "fused" / "autograd"; tce_agent.CriticPlan)."""
        plan = CriticPlan(self._critic_path() or "autograd",
                          int(self.num_minibatchs or 1),
                          bool(self.overlap_updates), bool(self.lazy_metrics))
        self.last_critic_plan = plan
        return plan

    def policy_plan(self, dataset, states=None):
        """kind: "smlp" / "direct" / "autograd" (tce_agent.PolicyPlan)."""
        plan = PolicyPlan(self._policy_path(dataset) or "autograd",
                          self._balance_iteration(), bool(self.graph_epochs))
        self.last_policy_plan = plan
        return plan

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
