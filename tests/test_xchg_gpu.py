"""The in-library one-shot gradient exchange (csrc/xchg.h, include/tce_hip.h
tce_xchg_*; SURVEY 8e -- no reference counterpart, mprl/ has no collectives):

* two / three PROCESSES on the one GPU, buffers mapped through
  hipIpcGetMemHandle / hipIpcOpenMemHandle, handles exchanged over gloo: the
  all-reduce is the rank-ordered sum, bit for bit, on every rank (and ==
  dist.all_reduce with two ranks: a + b in either order); exchange + Adam in
  one launch == all-reduce, then tce_adam_once_*;
* a peer that never arrives: the kernel ends at the limit and the status word
  names the peer (no hang);
(Ranks are processes here, as in a job: the streams of ONE process share
hardware queues, so a rank's kernel can queue behind a waiting one.)
"""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rank_ordered_sum(parts):
    s = parts[0].clone()
    for p in parts[1:]:
        s = s + p
    return s


def test_a_missing_peer_ends_at_the_limit_and_is_named():
    from tce_rl_amd.dist import Exchange
    xs = [Exchange(r, 2, 1 << 16) for r in range(2)]
    try:
        xs[0].connect_local(xs[1])
        xs[0].set_timeout_ms(100)
        t = torch.ones(5000, device="cuda")
        xs[0].allreduce(t)                  # rank 1 never launches
        torch.cuda.synchronize()            # returns: the wait is bounded
        assert xs[0].status() == 2          # 1 + the rank waited for
        with pytest.raises(RuntimeError, match="rank 0 waited.*rank 1"):
            xs[0].check()
        with pytest.raises(RuntimeError, match="does not fit"):
            xs[0].allreduce(torch.ones(1 << 16, device="cuda"))
    finally:
        for x in xs:
            x.close()


def test_one_rank_exchange_is_the_identity():
    from tce_rl_amd.dist import Exchange
    x = Exchange(0, 1, 1 << 16)
    try:
        t = torch.randn(9000, device="cuda")
        ref = t.clone()
        x.allreduce(t)
        torch.cuda.synchronize()
        assert torch.equal(t, ref) and x.status() == 0
    finally:
        x.close()


def _ipc_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port,
                            rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from tce_rl_amd.dist import Exchange
    from tce_rl_amd.optim import FlatAdam
    x = Exchange.over_group(1 << 20)
    ok = x is not None
    bad = []
    if ok:
        g = torch.Generator(device="cuda").manual_seed(17 + rank)

        def gathered_sum(a):
            parts = [torch.empty_like(a) for _ in range(world)]
            dist.all_gather(parts, a)
            return _rank_ordered_sum(parts)
        n0 = x.counters()[0]                 # (the self-test's collectives)
        for dtype in (torch.float32, torch.float64):
            # sizes around the workgroup granularity (4096 elements); three in
            # a row on one size: both slots and the reuse of the first
            for n in (1, 63, 4096, 4097, 28161, 100000):
                for _ in range(3):
                    a = torch.randn(n, dtype=dtype, device="cuda", generator=g)
                    want = gathered_sum(a)
                    b = a.clone()
                    x.allreduce(a)
                    torch.cuda.synchronize()
                    if not torch.equal(a, want):
                        bad.append(("allreduce", str(dtype), n))
                    if world == 2:      # a + b in either order: gloo's sum too
                        dist.all_reduce(b)
                        if not torch.equal(a, b):
                            bad.append(("vs dist.all_reduce", str(dtype), n))
        if x.counters()[0] - n0 != 2 * 6 * 3:
            bad.append(("counters", x.counters()))
        # exchange + Adam in one launch == all-reduce, then tce_adam_once_*
        for dtype in (torch.float32, torch.float64):
            for clip in (0.0, 0.5):
                n = 22657
                gw = torch.Generator(device="cuda").manual_seed(5)
                w0 = torch.randn(n, dtype=dtype, device="cuda", generator=gw)
                mk = lambda: FlatAdam([torch.nn.Parameter(w0.clone())], lr=3e-3,
                                      weight_decay=1e-4)
                opt, ref = mk(), mk()
                norms = torch.zeros(2, dtype=dtype, device="cuda")
                for step in range(3):
                    gr = torch.randn(n, dtype=dtype, device="cuda", generator=g)
                    ref.flat_grad.copy_(gathered_sum(gr))
                    ref.step_once(clip, grad_scale=1.0 / world)
                    opt.flat_grad.copy_(gr)
                    opt.step_exchange(x, clip, 1.0 / world, norms)
                    torch.cuda.synchronize()
                    same = (torch.equal(opt.flat_grad, ref.flat_grad)  # the sum stays
                            and torch.equal(opt.flat_param, ref.flat_param)
                            and torch.equal(opt.m, ref.m)
                            and torch.equal(opt.v, ref.v)
                            and float(opt.dev_state[0]) == step + 1)
                    # (the norm is summed in another order: workgroup partials)
                    close = torch.allclose(norms, ref.dev_state[1:3], rtol=1e-5,
                                           atol=0)
                    if not (same and close):
                        bad.append(("adam", str(dtype), clip, step, close,
                                    [float((a - b).abs().max()) for a, b in (
                                        (opt.flat_grad, ref.flat_grad),
                                        (opt.flat_param, ref.flat_param),
                                        (opt.m, ref.m), (opt.v, ref.v))],
                                    float(opt.dev_state[0])))
        st = x.status()
        x.close()
    else:
        st = -1
    q.put((rank, ok, bad, st))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_processes_on_one_gpu_through_hip_ipc(world):
    """The one-shot protocol across PROCESSES: buffers shared by
    hipIpcGetMemHandle / hipIpcOpenMemHandle, handles exchanged over gloo, all
    ranks on cuda:0.  The all-reduce == the rank-ordered sum on every rank, bit
    for bit (and == dist.all_reduce with two ranks); exchange + Adam in one
    launch == all-reduce, then tce_adam_once_*."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29850 + (os.getpid() % 100) + 7 * world
    procs = [ctx.Process(target=_ipc_worker, args=(r, world, port, q))
             for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, bad, st in out:
        assert ok, "the exchange's self-test failed on rank %d" % rank
        assert not bad, (rank, bad[:5])
        assert st == 0
