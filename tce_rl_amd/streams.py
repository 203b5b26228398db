"""Process-wide side streams.

A HIP stream is bound to one of the device's few hardware queues when it is
created (round robin), and streams that share a hardware queue wait for each
other's kernels.  An agent that created its own second stream therefore
behaved differently depending on how many streams the process had created
before it (measured in bench.py's `configs` block: the same table-tennis step
274 ms in a fresh process, 347 ms as the fifth agent of one process).  Every
agent of a process uses THE policy stream, THE graph stream of its device:
the mapping is that of the first agent, in every process.
"""
import torch

_STREAMS = {}


def get(name, device=None):
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None \
        else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = (name, dev.index)
    s = _STREAMS.get(key)
    if s is None:
        s = _STREAMS[key] = torch.cuda.Stream(device=dev)
    return s
