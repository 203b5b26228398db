"""World-size-2 gloo tests (CPU tensors) of the env-shard exchange step: flat
gradient all-reduce, merged advantage statistics, averaged scalars.  The HIP
kernels are not involved: this is the collective plumbing of tce_rl_amd/dist.py
and ops.merge_stats."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tce_rl_amd.dist import DistContext
    from tce_rl_amd import ops
    try:
        ctx = DistContext()
        assert ctx.world == world and ctx.rank == rank
        # --- gradient exchange: mean of per-shard mean-loss gradients equals
        # the gradient of the global mean loss (equal shards)
        torch.manual_seed(0)
        X = torch.randn(8, 5, dtype=torch.float64)
        y = torch.randn(8, dtype=torch.float64)
        lin = torch.nn.Linear(5, 1, dtype=torch.float64)
        ctx.broadcast_params(list(lin.parameters()))
        sl = slice(rank * 4, rank * 4 + 4)
        loss = (lin(X[sl]).squeeze(-1) - y[sl]).pow(2).mean()
        loss.backward()
        params = list(lin.parameters())
        ctx.allreduce_grads(params)
        ref = torch.nn.Linear(5, 1, dtype=torch.float64)
        ref.load_state_dict(lin.state_dict())
        (ref(X).squeeze(-1) - y).pow(2).mean().backward()
        for p, r in zip(params, ref.parameters()):
            assert torch.allclose(p.grad, r.grad, rtol=1e-12, atol=1e-14)
        # --- merged statistics == statistics of the global batch
        g = torch.Generator().manual_seed(1)
        full = torch.randn(10, generator=g, dtype=torch.float64) * 3 + 1
        mine = full[:3] if rank == 0 else full[3:]          # unequal shards
        n = torch.tensor(float(mine.numel()), dtype=torch.float64)
        stats = torch.stack([n, mine.mean(),
                             ((mine - mine.mean()) ** 2).sum()])
        merged = ops.merge_stats(stats)
        assert merged[0].item() == 10
        assert torch.allclose(merged[1], full.mean())
        assert torch.allclose((merged[2] / (merged[0] - 1)).sqrt(), full.std())
        # --- averaged scalar (initial entropy)
        m = ctx.mean_scalar(torch.tensor(float(rank + 1), dtype=torch.float64))
        assert m.item() == 1.5
        q.put((rank, "ok"))
    except Exception as e:                                   # noqa: BLE001
        q.put((rank, "FAIL: %r" % (e,)))
    finally:
        dist.destroy_process_group()


def test_env_shard_exchange_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q))
             for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(msg == "ok" for _, msg in res), res
