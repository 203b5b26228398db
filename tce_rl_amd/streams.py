"""Process-wide side streams, chosen by PROBING.

A HIP stream is bound to one of the device's few hardware queues when it is
created (round robin over every stream the process creates: torch's pool, the
communicators', the library's), and streams that share a hardware queue wait
for each other's kernels.  Which queue a stream lands on therefore depends on
how many streams the process created before it -- round 3 saw the same
table-tennis step take 274 ms in a fresh process and 347 ms as the fifth agent
of one process; round 5 saw the policy stream of a sharded run share the main
stream's queue once the communicators no longer created theirs first (the
policy epochs then START when the critic's first launches have ended: 125
instead of 107 ms per C2 step).

So a side stream is not taken on trust: ``get`` asks torch for streams until
one demonstrably runs BESIDE the default stream and every side stream handed
out before -- a 300 us spin kernel (tce_spin_us) goes on the other stream, a
5 us one on the candidate, and the candidate's must finish while the other is
still spinning.  One-off cost: a few hundred microseconds per stream and
process.  Every agent of a process uses THE policy stream, THE graph stream of
its device, so the mapping is that of the first agent.
"""
import warnings

import torch

_STREAMS = {}
PROBE_LOG = []          # (name, candidates tried, found) per stream: for tests / debugging


def _beside(cand, other):
    """Does a kernel on `cand` run while one on `other` is still running?"""
    from ._lib import call
    for _ in range(3):                    # (a preempted host can fake a "no")
        e_other, e_cand = torch.cuda.Event(), torch.cuda.Event()
        call("tce_spin_us", 300.0, other.cuda_stream)
        e_other.record(other)
        call("tce_spin_us", 5.0, cand.cuda_stream)
        e_cand.record(cand)
        e_cand.synchronize()
        beside = not e_other.query()
        e_other.synchronize()
        if beside:
            return True
    return False


def get(name, device=None):
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None \
        else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = (name, dev.index)
    s = _STREAMS.get(key)
    if s is not None:
        return s
    others = [torch.cuda.default_stream(dev)] + \
        [v for (n, i), v in _STREAMS.items() if i == dev.index]
    cur = torch.cuda.current_stream(dev)
    if all(cur.cuda_stream != o.cuda_stream for o in others):
        others.append(cur)
    torch.cuda.synchronize(dev)           # idle queues: the probe measures nothing else
    found, tried = None, 0
    with torch.cuda.device(dev):
        for tried in range(1, 33):
            cand = torch.cuda.Stream(device=dev)
            if all(_beside(cand, o) for o in others):
                found = cand
                break
    PROBE_LOG.append((name, tried, found is not None))
    if found is None:
        warnings.warn(
            "tce_rl_amd.streams: no stream of 32 runs beside the %d streams "
            "already in use on %s (hardware queues exhausted): the %r work "
            "will wait for kernels of another stream" % (len(others), dev, name),
            RuntimeWarning)
        found = cand
    _STREAMS[key] = found
    return found


def objective_side(device=None):
    """The fused objective's second stream (csrc/objective.hip), probed like the
    others and handed to the library."""
    from ._lib import call
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None \
        else torch.device(device)
    key = ("objective", dev.index if dev.index is not None
           else torch.cuda.current_device())
    fresh = key not in _STREAMS
    s = get("objective", dev)
    if fresh:
        call("tce_policy_objective_use_stream", s.cuda_stream)
    return s
