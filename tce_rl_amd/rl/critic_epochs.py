"""The critic update of both agents on the matrix-core epochs
(mprl/rl/agent/temporal_correlated_agent.py:323-379, black_box_agent.py:
105-157): full batch or minibatched, fused Adam, env shards."""
import numpy as np
import torch

from .. import util


class CriticEpochs:
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
        pinned buffers while the previous epoch runs.  "device": a keyed
        pseudo-random permutation computed on the GPU (tce_feistel_permutation:
        Feistel network + cycle walking, no sort, no library call) -- not the
        reference's sequence; for runs where the host draw (16 - 28 ms per epoch
        at 2 M rows) would be the step."""
        ag, n = self.agent, self.n_rows
        if getattr(ag, "minibatch_permutation", "numpy") == "device":
            from .._lib import call, ptr, stream
            # (the key from numpy's global generator: ONE draw per epoch where
            # the reference's shuffle makes n; seeding it seeds the pieces)
            key = int(np.random.randint(0, 2 ** 63 - 1, dtype=np.int64))
            out = torch.empty(n, dtype=torch.int64, device=ag.device)
            call("tce_feistel_permutation", ptr(out), n, key, stream())
            return out
        ring = ag.__dict__.setdefault("_perm_ring", [])
        if len(ring) < 2 or ring[0][0].numel() != n:
            if ring and ring[0][0].numel() != n:
                ring.clear()
            host = torch.empty(n, dtype=torch.int64).pin_memory()
            ring.append([host, None])
        slot = ring[0]
        ring.reverse()
        # np.random.shuffle of arange(n): the reference's draw, the same
        # generator calls.  Shuffled in ordinary memory (in the pinned buffer
        # itself the shuffle ran 40 % slower), then one copy into the slot
        idx = np.arange(n)
        np.random.shuffle(idx)
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
