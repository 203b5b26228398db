"""Host time per all-reduce call through torch.distributed.all_reduce against the
process group's own allreduce (cached options), one-rank RCCL world, 88 KB buffer.
    python scripts/time_allreduce_host.py"""
import os, time
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29741")
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
t = torch.zeros(22000, device="cuda")
pg = dist.distributed_c10d._get_default_group()
opts = dist.AllreduceOptions()
opts.reduceOp = dist.ReduceOp.SUM
for name, fn in (("dist.all_reduce", lambda: dist.all_reduce(t)),
                 ("pg.allreduce(cached opts).wait()", lambda: pg.allreduce([t], opts).wait()),
                 ("dist.all_reduce", lambda: dist.all_reduce(t))):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        fn()
    host = (time.perf_counter() - t0) / 2000 * 1e6
    torch.cuda.synchronize()
    dev = (time.perf_counter() - t0) / 2000 * 1e6
    print("%-36s host %.1f us per call, device-complete %.1f us per call" % (name, host, dev), flush=True)
dist.destroy_process_group()
