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

The gradient exchange itself lives in the library (``Exchange`` below,
csrc/xchg.h): the kernels that finish an epoch publish their gradient in a
peer-visible buffer, add the peers' values over xGMI in rank order and apply
Adam -- a sharded epoch stays ONE C call, as the un-sharded one.  The
``torch.distributed`` all-reduce between two C calls remains selectable
(``TCE_EXCHANGE=rccl``) and is what a job falls back to -- loudly -- when the
ranks do not share a node or the start-up self-test of the exchange fails.
"""
import ctypes
import os
import warnings
import weakref

import torch
import torch.distributed as dist


# Collectives issued by this process since the last reset (count, payload
# bytes): bench.py reports them per step so that a reader of the scaling curve
# can see what the wire carried.
STATS = {"collectives": 0, "bytes": 0}
_LIVE = weakref.WeakSet()           # the Exchange objects of this process


def stats():
    """Collectives issued by this process since the last reset: the
    torch.distributed ones counted here plus those the library issued inside
    its kernels (every live Exchange's counters)."""
    for x in list(_LIVE):
        x.drain_counters()
    return dict(STATS)


def reset_stats():
    stats()
    STATS["collectives"] = STATS["bytes"] = 0


def _count(t):
    STATS["collectives"] += 1
    STATS["bytes"] += t.numel() * t.element_size()


# The small per-step collectives (statistics gathers, pair broadcast, the critic
# split's MAX, column sums) ride on a third in-library exchange when one exists
# ("aux", set up by the agent next to the two gradient exchanges): a sum over
# ranks of a float64 image of the tensor -- a gather is the sum of images that
# are zero outside the own row, a broadcast the sum of rank 0's tensor and
# zeros -- so a steady-state step issues NO torch.distributed call and no
# communicator stream is ever created by it.  The aux collectives are issued
# from whatever stream is current; an event chain keeps them in stream order
# (csrc/xchg.h: one exchange, one stream order on every rank).
# One aux exchange per set of ranks, shared (reference-counted) by every
# DistContext over those ranks: a process that builds several agents under one
# process group (bench `configs`, the tests) neither leaks a buffer + W - 1 IPC
# mappings per agent nor has an earlier agent silently ride on a later agent's
# exchange.
_AUX = {}                           # ranks tuple -> {"x", "event", "refs"}
_AUX_ELEMS = 1 << 16


def _group_key(group=None):
    """The ranks of `group` (None: the default group) as a tuple."""
    if group is None:
        return tuple(range(dist.get_world_size()))
    return tuple(dist.get_process_group_ranks(group))


def _aux_slot(group):
    if not _AUX or not (dist.is_available() and dist.is_initialized()):
        return None
    return _AUX.get(_group_key(group))


def _aux_for(t, group):
    slot = _aux_slot(group)
    if slot is None or not t.is_cuda:
        return None
    x = slot["x"]
    return x if x is not None and x.handle else None


def _aux_gather(x, t):
    """[world, n] float64 = every rank's t (flattened): ONE launch."""
    import torch
    cur = torch.cuda.current_stream()
    if x.order_event is not None:
        cur.wait_event(x.order_event)
    mine = t.reshape(-1).to(torch.float64).contiguous()
    out = torch.empty(x.world, mine.numel(), dtype=torch.float64,
                      device=t.device)
    x.allgather(mine, out)
    ev = torch.cuda.Event()
    ev.record(cur)
    x.order_event = ev
    return out


def _aux_sum(x, buf):
    """In-place rank-ordered sum of a contiguous float64 device tensor."""
    import torch
    cur = torch.cuda.current_stream()
    if x.order_event is not None:
        cur.wait_event(x.order_event)
    flat = buf.reshape(-1)
    for i in range(0, flat.numel(), _AUX_ELEMS):
        x.allreduce(flat[i:i + _AUX_ELEMS])
    ev = torch.cuda.Event()
    ev.record(cur)
    x.order_event = ev


def all_reduce(t, op=None, group=None):
    import torch
    x = _aux_for(t, group)
    if x is not None and op in (None, dist.ReduceOp.SUM, dist.ReduceOp.MAX):
        if op == dist.ReduceOp.MAX:
            t.copy_(_aux_gather(x, t).amax(0).reshape(t.shape))
        elif t.numel() <= 4096:
            # (small: gather + a sum in rank order, one exchange launch)
            t.copy_(_aux_gather(x, t).sum(0).reshape(t.shape))
        else:
            buf = t.to(torch.float64).contiguous()
            if buf.data_ptr() == t.data_ptr():
                _aux_sum(x, t)
            else:
                _aux_sum(x, buf)
                t.copy_(buf)
        return
    _count(t)
    dist.all_reduce(t, op=dist.ReduceOp.SUM if op is None else op, group=group)


def all_gather_into_tensor(out, t, group=None):
    import torch
    x = _aux_for(t, group)
    if x is not None and t.numel() <= (1 << 16):
        rows = _aux_gather(x, t)
        if out.dtype == torch.float64 and out.is_contiguous():
            out.copy_(rows.reshape(out.shape))       # (same dtype: a plain copy)
        else:
            out.copy_(rows.reshape(out.shape))
        return
    _count(out)
    dist.all_gather_into_tensor(out, t, group=group)


def broadcast(t, src=0, group=None):
    import torch
    x = _aux_for(t, group)
    if x is not None and t.numel() <= 4096:
        t.copy_(_aux_gather(x, t)[src].reshape(t.shape))
        return
    if x is not None:
        buf = t.to(torch.float64).contiguous() if x.rank == src else \
            torch.zeros(t.shape, dtype=torch.float64, device=t.device)
        if buf.data_ptr() == t.data_ptr():
            buf = buf.clone()
        _aux_sum(x, buf)
        t.copy_(buf)
        return
    _count(t)
    dist.broadcast(t, src=src, group=group)


class Exchange:
    """This rank's end of a one-shot in-library exchange (include/tce_hip.h:
    tce_xchg_*).  ``handle`` is what the ``xchg`` argument of the epoch calls
    takes.  Every rank must issue the same collectives on it, from one
    stream."""

    def __init__(self, rank, world, max_bytes):
        from . import _lib
        self._lib = _lib.load()
        self.rank, self.world, self.max_bytes = rank, world, int(max_bytes)
        h = ctypes.c_void_p()
        if self._lib.tce_xchg_create(rank, world, self.max_bytes,
                                     ctypes.byref(h)):
            raise RuntimeError("tce_xchg_create: " +
                               self._lib.tce_last_error().decode())
        self.handle = h.value
        self._counted = (0, 0)
        # the last collective issued on this exchange (one exchange = one
        # stream order on every rank: issuers on other streams wait for it)
        self.order_event = None
        self.self_test = None           # True / False once over_group has run it
        _LIVE.add(self)

    def export(self):
        buf = ctypes.create_string_buffer(self._lib.tce_xchg_handle_bytes())
        self._call("tce_xchg_export", self.handle, buf)
        return buf.raw

    def connect(self, handles):
        """handles: the ``export()`` of every rank, in rank order."""
        assert len(handles) == self.world
        self._call("tce_xchg_connect", self.handle, b"".join(handles))

    def connect_local(self, peer):
        """A peer rank that lives in this process (tests)."""
        self._call("tce_xchg_connect_local", self.handle, peer.rank,
                   peer.handle)

    def _call(self, name, *args):
        if getattr(self._lib, name)(*args):
            raise RuntimeError("%s: %s" % (
                name, self._lib.tce_last_error().decode()))

    def allreduce(self, t):
        """In-place sum over ranks (rank order) of a contiguous device tensor."""
        from ._lib import sfx, stream
        assert t.is_contiguous() and t.is_cuda
        self._call("tce_xchg_allreduce_" + sfx(t.dtype), self.handle,
                   t.data_ptr(), t.numel(), stream())

    def allgather(self, mine, out):
        """out [world, n] = every rank's mine [n] (contiguous float64 device
        tensors; one launch)."""
        from ._lib import stream
        import torch
        assert mine.dtype == out.dtype == torch.float64 and mine.is_cuda \
            and mine.is_contiguous() and out.is_contiguous() \
            and out.numel() == self.world * mine.numel()
        self._call("tce_xchg_allgather_f64", self.handle, mine.data_ptr(),
                   out.data_ptr(), mine.numel(), stream())

    def status(self):
        return int(self._lib.tce_xchg_status(self.handle))

    def check(self):
        """Raise if a wait for a peer ran into the limit (the kernels go on
        instead of hanging; what they computed since is not to be used)."""
        st = self.status()
        if st:
            raise RuntimeError(
                "gradient exchange: rank %d waited longer than the limit "
                "(TCE_XCHG_TIMEOUT_MS) for rank %d -- a peer died or fell out "
                "of step" % (self.rank, st - 1))

    def counters(self):
        n, b = ctypes.c_int64(), ctypes.c_int64()
        self._call("tce_xchg_counters", self.handle, ctypes.byref(n),
                   ctypes.byref(b))
        return n.value, b.value

    def drain_counters(self):
        """Add the collectives issued since the last call to STATS."""
        if not self.handle:
            return
        n, b = self.counters()
        STATS["collectives"] += n - self._counted[0]
        STATS["bytes"] += b - self._counted[1]
        self._counted = (n, b)

    def wait_stats(self, reset=False):
        """(total_us, max_us, collectives): how long workgroup 0 of this rank's
        collectives waited for its slowest peer (device-side clock, csrc/xchg.h).
        Blocking."""
        tot, mx, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        self._call("tce_xchg_wait_stats", self.handle, ctypes.byref(tot),
                   ctypes.byref(mx), ctypes.byref(n), int(bool(reset)))
        return tot.value, mx.value, n.value

    def set_timeout_ms(self, ms):
        self._call("tce_xchg_set_timeout_ms", self.handle, float(ms))

    def close(self):
        if self.handle:
            self._lib.tce_xchg_destroy(self.handle)
            self.handle = None

    @classmethod
    def over_group(cls, max_bytes, group=None):
        """Collective: every rank of `group` creates its end, the IPC handles
        travel through an all-gather, and a self-test (three all-reduces of
        known integers, checked on every rank, agreed with a MIN all-reduce)
        decides whether the exchange is used.  Returns None -- on EVERY rank --
        when the ranks span hosts or the test fails."""
        import socket
        import torch
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        if world > 8:
            return None
        # (a rank whose own end cannot be created must still take part in the
        # collectives below, or its peers would wait for it forever)
        x, handle = None, None
        try:
            x = cls(rank, world, max_bytes)
            handle = x.export() if world > 1 else None
        except RuntimeError as e:
            warnings.warn("gradient exchange: creating the peer-visible buffer "
                          "failed (%s)" % e)
            if world == 1:
                return None
        ok = True
        if world > 1:
            mine = (socket.gethostname(), handle)
            everyone = [None] * world
            # (host-side bootstrap: a gloo twin of the group, so that no
            # communicator kernel -- and no communicator stream -- is needed)
            dist.all_gather_object(everyone, mine, group=_boot_group(group))
            ok = all(h == mine[0] and b is not None for h, b in everyone)
            if ok:
                try:
                    x.connect([b for _, b in everyone])
                except RuntimeError as e:
                    warnings.warn("gradient exchange: mapping the peers failed "
                                  "(%s)" % e)
                    ok = False
            ok = _agree(ok, group)
            if ok:
                # a short limit while probing: a mapping that does not show the
                # peers' stores costs seconds here, not six times the default
                x.set_timeout_ms(float(os.environ.get(
                    "TCE_XCHG_SELFTEST_TIMEOUT_MS", "5000")))
                x.self_test = x._self_test()
                ok = _agree(x.self_test, group)
                if ok:
                    x.set_timeout_ms(float(os.environ.get(
                        "TCE_XCHG_TIMEOUT_MS", "20000")))
        if not ok:
            if world > 1 and rank == 0:
                warnings.warn(
                    "gradient exchange: the in-library xGMI exchange is not "
                    "usable on this job (ranks on several hosts, IPC mapping "
                    "or self-test failed); falling back to "
                    "torch.distributed all-reduces between the C calls")
            if x is not None:
                x.close()
            return None
        return x

    def _self_test(self):
        import torch
        dev = torch.device("cuda", torch.cuda.current_device())
        good = True
        for dtype in (torch.float32, torch.float64):
            cap = self.max_bytes // (4 if dtype == torch.float32 else 8)
            for n in (1, min(cap, 4097), min(cap, 70001)):
                base = (torch.arange(n, device=dev) % 1021).to(dtype)
                t = base * (self.rank + 1)
                self.allreduce(t)
                want = base * (self.world * (self.world + 1) // 2)
                good = good and bool(torch.equal(t, want))
        torch.cuda.synchronize()
        return good and self.status() == 0


_BOOT = {}


def _boot_group(group=None):
    """A gloo group over the ranks of `group` (collective on first use) for the
    host-side bootstrap; the group itself when it is gloo already."""
    if dist.get_backend(group) == "gloo":
        return group
    key = _group_key(group)
    if key not in _BOOT:
        # dist.new_group is collective over the DEFAULT group: every process of
        # the job must reach this point.  DistContext creates the twin eagerly
        # in its constructor and accepts only groups that span the whole job,
        # so the lazy callers below always find it.
        if len(key) != dist.get_world_size():
            raise ValueError(
                "tce_rl_amd.dist: a process group over a strict subset of the "
                "job's ranks is not supported (the host-side gloo twin is "
                "created collectively over the default group)")
        _BOOT[key] = dist.new_group(ranks=list(key), backend="gloo")
    return _BOOT[key]


def _agree(flag, group=None):
    """True on every rank iff `flag` is true on every rank."""
    import torch
    t = torch.tensor([1 if flag else 0], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=_boot_group(group))
    return bool(t.item())


def host_barrier(group=None):
    """A barrier that launches nothing on the device (gloo twin of the group)."""
    dist.barrier(group=_boot_group(group))


def exchange_wanted():
    """TCE_EXCHANGE: "xgmi" (default: the in-library one-shot exchange) or
    "rccl" (torch.distributed all-reduces between the C calls)."""
    mode = os.environ.get("TCE_EXCHANGE", "xgmi").lower()
    if mode not in ("xgmi", "rccl"):
        raise ValueError("TCE_EXCHANGE=%r (xgmi | rccl)" % mode)
    return mode == "xgmi"


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
        self._exchanges = {}
        self._owns_aux = False
        self._closed = False
        if self.enabled:
            # collective contract: a DistContext (an agent) is constructed by
            # every rank of the job at the same point
            _boot_group(group)

    def exchange(self, channel, max_bytes):
        """The in-library exchange of one update chain ("critic" / "policy";
        two chains run on two streams, so each has its own).  Collective on
        first use; None when the torch.distributed path is to be taken."""
        if not self.active or not exchange_wanted():
            return None
        if channel not in self._exchanges:
            import torch
            if not torch.cuda.is_available():
                self._exchanges[channel] = None
            else:
                self._exchanges[channel] = Exchange.over_group(max_bytes,
                                                               self.group)
        return self._exchanges[channel]

    def setup_aux(self):
        """The exchange of the small per-step collectives (collective call)."""
        # (TCE_AUX_EXCHANGE=0: keep those on torch.distributed, for A / B runs)
        if os.environ.get("TCE_AUX_EXCHANGE", "1") == "0":
            return None
        if not self.active or not exchange_wanted():
            return None
        key = _group_key(self.group)
        slot = _AUX.get(key)
        if slot is None or slot["x"] is None or not slot["x"].handle:
            import torch
            x = Exchange.over_group(8 * _AUX_ELEMS, self.group) \
                if torch.cuda.is_available() else None
            if x is None:
                return None
            slot = _AUX[key] = {"x": x, "refs": 0}
        if not self._owns_aux:
            slot["refs"] += 1
            self._owns_aux = True
        self._exchanges["aux"] = slot["x"]
        return slot["x"]

    def close(self):
        """Collective: host barrier (no peer may still be reading this rank's
        buffers), then the gradient exchanges of this context are destroyed and
        its share of the aux exchange released.  The agents call it from
        ``close()``; bench.py before ``destroy_process_group``."""
        if self._closed:
            return
        self._closed = True
        live = [x for k, x in self._exchanges.items()
                if x is not None and k != "aux"]
        aux_last = False
        if self._owns_aux:
            slot = _AUX.get(_group_key(self.group))
            if slot is not None:
                slot["refs"] -= 1
                aux_last = slot["refs"] <= 0
        if (live or aux_last) and self.enabled and dist.is_initialized():
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            host_barrier(self.group)
        for x in live:
            x.close()
        if aux_last:
            slot = _AUX.pop(_group_key(self.group), None)
            if slot is not None and slot["x"] is not None:
                slot["x"].close()
        self._exchanges = {}
        self._owns_aux = False

    def exchange_report(self):
        """Per channel: kind, self-test result, wait telemetry (blocking)."""
        out = {}
        for k, x in self._exchanges.items():
            if x is None:
                out[k] = {"kind": "rccl"}
                continue
            tot, mx, n = x.wait_stats()
            out[k] = {"kind": "xgmi-oneshot", "self_test": x.self_test,
                      "collectives": n,
                      "wait_us_mean": round(tot / n, 2) if n else None,
                      "wait_us_max": round(mx, 2)}
        return out

    def reset_wait_stats(self):
        for x in self._exchanges.values():
            if x is not None:
                x.wait_stats(reset=True)

    def exchange_kind(self):
        """What carries the gradients: "xgmi-oneshot" | "rccl" | "none"."""
        if not self.active:
            return "none"
        xs = [x for x in self._exchanges.values()]
        return "xgmi-oneshot" if xs and all(x is not None for x in xs) \
            else "rccl"

    def check_exchanges(self):
        for x in self._exchanges.values():
            if x is not None:
                x.check()
                x.drain_counters()

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
