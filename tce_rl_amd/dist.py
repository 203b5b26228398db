"""Env-sharded data parallelism over RCCL (torch.distributed, one process per
GPU).  The reference is single-device (no collectives anywhere under mprl/);
this is the new exchange step of the sharded path:

* one flat all-reduce (sum) of the gradients per optimizer step, divided by the
  world size -- policy + critic are replicated, every rank holds N/world envs,
  local losses are means over the local shard, so the result equals the
  reference's global-batch mean gradient;
* merged (count, mean, M2) statistics for advantage normalisation
  (ops.merge_stats) and a mean of the initial entropy.

Messages are tiny (<= ~300 KB): latency-bound, so exactly one collective per
step on one flat buffer.
"""
import os

import torch
import torch.distributed as dist


# Collectives issued by this process since the last reset (count, payload
# bytes): bench.py reports them per step so that a reader of the scaling curve
# can see what the wire carried.
STATS = {"collectives": 0, "bytes": 0}


def reset_stats():
    STATS["collectives"] = STATS["bytes"] = 0


def _count(t):
    STATS["collectives"] += 1
    STATS["bytes"] += t.numel() * t.element_size()


def all_reduce(t, op=None, group=None):
    _count(t)
    dist.all_reduce(t, op=dist.ReduceOp.SUM if op is None else op, group=group)


def all_gather_into_tensor(out, t, group=None):
    _count(out)
    dist.all_gather_into_tensor(out, t, group=group)


def broadcast(t, src=0, group=None):
    _count(t)
    dist.broadcast(t, src=src, group=group)


def active(group=None):
    """True when the sharded code path (collectives included) is to be taken:
    a process group of more than one rank, or -- TCE_FORCE_DIST=1 -- any
    initialised process group, so that a ONE-rank RCCL world on a one-GPU box
    runs exactly the launches and collectives of an N-rank job."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or \
        os.environ.get("TCE_FORCE_DIST") == "1"


class DistContext:
    def __init__(self, group=None):
        self.group = group
        self.enabled = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.rank = dist.get_rank(group) if self.enabled else 0
        self.active = active(group)
        self._flat = {}
        self._aux_group = None

    def aux_group(self):
        """A second communicator over the same ranks (collective call: every
        rank must reach it at the same point).  The overlapped update issues
        the critic's and the policy's gradient all-reduces from two streams;
        on one communicator they would be serialised in host issue order (all
        critic epochs first), on two they proceed independently."""
        if not self.active:
            return self.group
        if self._aux_group is None:
            ranks = dist.get_process_group_ranks(self.group) \
                if self.group is not None else list(range(self.world))
            # no fallback: a failure on SOME ranks would leave them on
            # different communicators (dead-lock), so it has to surface
            self._aux_group = dist.new_group(ranks=ranks)
        return self._aux_group

    def allreduce_grads(self, params):
        if not self.active:
            return
        grads = [p.grad for p in params]
        key = id(params)
        n = sum(g.numel() for g in grads)
        flat = self._flat.get(key)
        if flat is None or flat.numel() != n or flat.device != grads[0].device:
            flat = torch.empty(n, dtype=grads[0].dtype, device=grads[0].device)
            self._flat[key] = flat
        torch.cat([g.reshape(-1) for g in grads], out=flat)
        all_reduce(flat, group=self.group)
        flat.div_(self.world)
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()

    def allreduce_flat(self, flat, group=None, average=True):
        """In-place sum (average=False) or mean over ranks of an already-flat
        gradient buffer."""
        if not self.active:
            return
        all_reduce(flat, group=self.group if group is None else group)
        if average:
            flat.div_(self.world)

    def mean_scalar(self, x):
        if not self.active:
            return x
        y = x.clone()
        all_reduce(y, group=self.group)
        return y / self.world

    def broadcast_params(self, params):
        if not self.active:
            return
        for p in params:
            broadcast(p.data, src=0, group=self.group)
