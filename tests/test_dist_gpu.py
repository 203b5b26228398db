"""Env-sharded data parallel path end to end on the GPU: two ranks (gloo
collectives on device tensors; both processes share the one GPU of the test
box) run agent.step() on different env shards and must stay in lock-step --
identical parameters after every optimizer step, finite metrics, global step
count = sum over ranks."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, overlap, q):
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port,
                            rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from tce_rl_amd.config import tce_config
    from tce_rl_amd.mp_exp import MPExperiment
    torch.manual_seed(100 + rank)            # different initial weights ...
    cfg = tce_config("metaworld", num_env=32, num_basis=5, epochs=3,
                     evaluation_interval=0, seed=rank)
    cfg["params"]["agent"]["args"]["overlap_updates"] = overlap
    exp = MPExperiment()
    exp.initialize(cfg, 0, None)             # ... made equal by the broadcast
    agent = exp.agent
    res = None
    pairs = []
    for _ in range(2):                       # ranks draw differently seeded
        res = agent.step()                   # pair offsets; rank 0's is used
        pairs.append(agent.sampler.pred_pairs.cpu().numpy().copy())
    flat = torch.cat([p.detach().reshape(-1).cpu()
                      for p in agent.policy.parameters + agent.critic.parameters])
    q.put((rank, flat.numpy(), float(res["critic_loss_mean"]),
           float(res["surrogate_loss_mean"]), int(res["num_global_steps"]),
           float(res["exploration_step_rewards_mean"]), pairs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
def test_two_ranks_stay_in_lock_step(overlap):
    import numpy as np
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 200) + (50 if overlap else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, q))
             for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, w0, c0, s0, g0, r0, p0), (_, w1, c1, s1, g1, r1, p1) = out
    assert all(np.array_equal(a, b) for a, b in zip(p0, p1))
    assert np.isfinite(w0).all() and np.isfinite([c0, c1, s0, s1]).all()
    assert np.array_equal(w0, w1)            # same parameters on both ranks
    assert g0 == g1 == 2 * 2 * 32 * 500      # iterations x ranks x envs x T
    assert r0 != r1                          # the shards really differ
