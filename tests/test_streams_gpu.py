"""tce_rl_amd/streams.py: side streams are PROBED -- a HIP stream is bound to a
hardware queue when it is created, streams that share a queue wait for each
other's kernels, and which queue a new stream lands on depends on how many
streams the process created before.  A side stream is handed out only if a
kernel on it demonstrably runs while one on the default stream (and on every
side stream handed out before) is still running."""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_spin_kernel_lasts_what_it_is_asked_to():
    from tce_rl_amd._lib import call, stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    call("tce_spin_us", 50.0, stream())              # warm
    torch.cuda.synchronize()
    e0.record()
    call("tce_spin_us", 2000.0, stream())
    e1.record()
    torch.cuda.synchronize()
    # at least what was asked for (the upper end is the launch's business: the
    # probe only compares two spins with each other)
    assert 1.9 <= e0.elapsed_time(e1) <= 200.0
    with pytest.raises(RuntimeError, match="spin_us"):
        call("tce_spin_us", 0.0, stream())


def test_side_streams_run_beside_the_default_stream_and_each_other():
    from tce_rl_amd import streams
    from tce_rl_amd._lib import call
    names = ("policy", "objective", "graph")
    got = [streams.get(n) for n in names]
    assert len({s.cuda_stream for s in got}) == 3
    assert streams.get("policy") is got[0]           # process-wide
    main = torch.cuda.default_stream()
    every = [main] + got
    for i, a in enumerate(every):
        for b in every[i + 1:]:
            assert streams._beside(b, a) and streams._beside(a, b)
    # the probe itself tells shared queues apart: a stream does not run beside
    # itself
    assert not streams._beside(got[0], got[0])
    # all three busy at once: the wall time is one spin, not three
    torch.cuda.synchronize()
    t = time.perf_counter()
    for s in got:
        call("tce_spin_us", 3000.0, s.cuda_stream)
    torch.cuda.synchronize()
    assert time.perf_counter() - t < 0.006
