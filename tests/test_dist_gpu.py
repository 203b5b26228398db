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


def _worker(rank, world, port, overlap, q, critic_arith="f32"):
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port,
                            rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from tce_rl_amd.config import tce_config
    from tce_rl_amd.mp_exp import MPExperiment
    torch.manual_seed(100 + rank)            # different initial weights ...
    # the GLOBAL env count: MPExperiment gives every rank 64 / world envs and
    # the env / noise seed `seed + rank`
    cfg = tce_config("metaworld", num_env=64, num_basis=5, epochs=3,
                     evaluation_interval=0, seed=0)
    cfg["params"]["agent"]["args"]["overlap_updates"] = overlap
    cfg["params"]["agent"]["args"]["critic_arith"] = critic_arith
    exp = MPExperiment()
    exp.initialize(cfg, 0, None)             # ... made equal by the broadcast
    agent = exp.agent
    assert agent.sampler.num_env_train == 32 and agent.sampler.seed == rank
    res = None
    pairs = []
    for _ in range(2):                       # ranks draw differently seeded
        res = agent.step()                   # pair offsets; rank 0's is used
        pairs.append(agent.sampler.pred_pairs.cpu().numpy().copy())
    flat = torch.cat([p.detach().reshape(-1).cpu()
                      for p in agent.policy.parameters + agent.critic.parameters])
    rms = agent.sampler.obs_rms
    flat = torch.cat([flat, rms.mean.detach().reshape(-1).cpu().float(),
                      rms.var.detach().reshape(-1).cpu().float(),
                      torch.tensor([float(rms.count)])])
    q.put((rank, flat.numpy(), float(res["critic_loss_mean"]),
           float(res["surrogate_loss_mean"]), int(res["num_global_steps"]),
           float(res["exploration_step_rewards_mean"]), pairs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap,critic_arith", [(False, "f32"), (True, "f32"),
                                                  (True, "bf16x3")])
def test_two_ranks_stay_in_lock_step(overlap, critic_arith):
    """(critic_arith bf16x3: the exchange behind csrc/mlpb.hip's slab reduction,
    tce_mlp_critic_bf16x3(..., xchg).)"""
    import numpy as np
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 200) + (50 if overlap else 0) + \
        (25 if critic_arith != "f32" else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, q, critic_arith))
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
    assert np.array_equal(w0, w1)            # same parameters AND obs statistics
    assert g0 == g1 == 2 * 2 * 32 * 500      # iterations x ranks x envs x T
    assert r0 != r1                          # the shards really differ


def _bbrl_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port,
                            rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from tce_rl_amd.config import bbrl_config
    from tce_rl_amd.mp_exp import MPExperiment
    torch.manual_seed(200 + rank)
    cfg = bbrl_config(num_env=128, epochs=4, seed=0)
    exp = MPExperiment()
    exp.initialize(cfg, 0, None)
    agent = exp.agent
    assert agent.sampler.num_env_train == 128 // world
    for _ in range(3):                       # iteration 1 carries the balance check
        res = agent.step()
    agent.flush_metrics()
    flat = torch.cat([p.detach().reshape(-1).cpu()
                      for p in agent.policy.parameters + agent.critic.parameters])
    q.put((rank, flat.numpy(), agent.dist.exchange_kind(),
           int(res["num_global_steps"]),
           float(res["exploration_episode_reward_mean"])))
    dist.barrier()
    dist.destroy_process_group()


def test_four_ranks_of_the_black_box_agent_stay_in_lock_step():
    """BASELINE configs[3] is a 4-GPU env shard: four ranks (processes on the one
    GPU of the test box, buffers mapped through HIP IPC) run BlackBoxAgent steps
    -- both update chains side by side, each with its own in-library exchange --
    and end with bit-identical replicas."""
    import numpy as np
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 100)
    procs = [ctx.Process(target=_bbrl_worker, args=(r, 4, port, q))
             for r in range(4)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=300) for _ in range(4)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w0 = out[0][1]
    assert np.isfinite(w0).all()
    for rank, w, kind, steps, rew in out:
        assert kind == "xgmi-oneshot"
        assert np.array_equal(w, w0), rank
        assert steps == 3 * 128 * 500
    assert len({o[4] for o in out}) == 4          # the shards really differ


def _rms_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port,
                            rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from tce_rl_amd import ops
    from tce_rl_amd.rl.sampler import RunningMeanStd
    g = torch.Generator().manual_seed(5)
    batches = [torch.randn(3000, 48, generator=g) * 3 + 1,
               torch.randn(1000, 48, generator=g) - 2]
    rms = RunningMeanStd(shape=(48,), dtype="torch.float32", device="cuda")
    for b in batches:                         # unequal shards on purpose
        cut = (2 * b.shape[0]) // 3
        rms.update((b[:cut] if rank == 0 else b[cut:]).cuda())
    out = [rms.mean.cpu(), rms.var.cpu(), torch.tensor([rms.count])]
    if rank == 0:                             # single-process result, same kernel
        mean = torch.zeros(48, device="cuda")
        var = torch.ones(48, device="cuda")
        count = 1e-4
        for b in batches:
            count = ops.rms_update(b.cuda(), mean, var, count)
        out += [mean.cpu(), var.cpu(), torch.tensor([count])]
    q.put((rank, [t.double().numpy() for t in out]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_obs_statistics_equal_the_single_process_ones():
    import numpy as np
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29850 + (os.getpid() % 100)
    procs = [ctx.Process(target=_rms_worker, args=(r, 2, port, q))
             for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    m0, v0, c0, m_ref, v_ref, c_ref = out[0]
    m1, v1, c1 = out[1]
    assert np.array_equal(m0, m1) and np.array_equal(v0, v1) and c0 == c1
    np.testing.assert_allclose(m0, m_ref, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(v0, v_ref, rtol=1e-5, atol=1e-6)
    assert c0[0] == pytest.approx(c_ref[0])


# ---------------------------------------------------------------------------
# sharded == single process (the only contract the new
# collectives have; reference sites whose global-batch means / stds they
# reproduce: temporal_correlated_agent.py:213-215,716,736)
# ---------------------------------------------------------------------------
N_GLOBAL = 64


def _forced(agent, dof, K, lo, hi):
    """Env state and parameter noise of the global batch rows [lo, hi)."""
    g = torch.Generator().manual_seed(7)
    goal = torch.rand(N_GLOBAL, dof, generator=g) * 2 - 1
    pos0 = 0.1 * (torch.rand(N_GLOBAL, dof, generator=g) * 2 - 1)
    eps = torch.randn(N_GLOBAL, K, generator=g)
    goal, pos0, eps = (t[lo:hi].cuda() for t in (goal, pos0, eps))
    env = agent.sampler.train_envs
    n = hi - lo

    def reset():
        env.goal = goal
        z = torch.zeros(n, dof, device="cuda")
        return env._obs(torch.zeros(n, device="cuda"), pos0, z)
    env.reset = reset
    sample = agent.policy.sample
    agent.policy.sample = lambda **kw: sample(**kw, eps=eps)


def _build_for_equivalence(kind, n_env, overlap):
    if kind.startswith("tce"):
        from tce_rl_amd.config import tce_config
        from tce_rl_amd.mp_exp import MPExperiment
        cfg = tce_config("metaworld", num_env=n_env, num_basis=5, epochs=3,
                         evaluation_interval=0, seed=0)
        cfg["params"]["agent"]["args"]["overlap_updates"] = overlap
        if kind == "tce_accrew":             # column means over the GLOBAL batch
            cfg["params"]["agent"]["args"]["segment_advantage"] = \
                "accumulated_rewards"
        exp = MPExperiment()
        exp.initialize(cfg, 0, None)
        return exp.agent, 4, 24
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from test_agent_gpu import BBRL_MID, build_bbrl
    if kind == "bbrl":
        agent, _ = build_bbrl(n_env, 3)
    else:
        # the reference's box-pushing / table-tennis black-box nets:
        # objective.BBDirectEpoch + the matrix-core / pmlp critic epochs
        c = BBRL_MID[kind[len("bbrl_"):]]
        agent, _ = build_bbrl(n_env, 3, c["policy_hidden"], c["critic_hidden"],
                              c["act"], c["std_only"], c["dtype"],
                              wd_policy=c["wd"], wd_critic=c["wd"])
        assert agent._critic_path() == (
            "pmlp" if c["critic_hidden"][1] == 1 else "fused")
    agent.evaluation_interval = 0
    return agent, 4, 20


def _equiv_run(kind, overlap, world, rank):
    """Two agent.step()s on the forced global batch (this rank's rows) from
    seed-0 initial weights -> flat vector {parameters, obs statistics,
    normalised segment advantages of the last step (this rank's rows)}."""
    torch.manual_seed(0)                     # identical initial weights
    n = N_GLOBAL // world
    # a sharded TCE experiment is configured with the GLOBAL env count
    agent, dof, K = _build_for_equivalence(
        kind, N_GLOBAL if kind.startswith("tce") else n, overlap)
    assert agent.sampler.num_env_train == n
    _forced(agent, dof, K, rank * n, (rank + 1) * n)
    captured = {}
    pd = agent.process_dataset

    def grab(ds):
        out = pd(ds)
        captured["adv"] = out["segment_advantage"].detach().cpu().reshape(n, -1)
        return out
    agent.process_dataset = grab
    bal = None
    for it in range(2):
        torch.manual_seed(20 + it)           # pair offsets (rank 0's are used)
        res = agent.step()
        if it == 0:
            # iteration 1 carries the policy balance check (1 in balance_check):
            # the norms of the surrogate's / the trust region loss's gradient
            # alone -- of the GLOBAL batch, however the envs are sharded
            bal = [float(res.get(k, float("nan"))) for k in
                   ("surrogate_grad_norm_mean", "trust_region_grad_norm_mean",
                    "balance_ratio")]
    flat = [p.detach().reshape(-1).cpu().double()
            for p in agent.policy.parameters + agent.critic.parameters]
    rms = getattr(agent.sampler, "obs_rms", None)
    if rms is not None:
        flat += [rms.mean.reshape(-1).cpu().double(),
                 rms.var.reshape(-1).cpu().double(),
                 torch.tensor([float(rms.count)], dtype=torch.float64)]
    return torch.cat(flat).numpy(), captured["adv"].double().numpy(), \
        int(res["num_global_steps"]), bal, agent.dist.exchange_kind()


def _equiv_worker(rank, world, port, kind, overlap, q):
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo",
                                init_method="tcp://127.0.0.1:%d" % port,
                                rank=rank, world_size=world)
    torch.cuda.set_device(0)
    q.put((world, rank) + _equiv_run(kind, overlap, world, rank))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("kind,overlap", [("tce", True), ("tce", False),
                                          ("tce_accrew", True),
                                          ("bbrl", False),
                                          ("bbrl_box_push", False),
                                          ("bbrl_table_tennis", False)])
def test_sharded_equals_single_process(kind, overlap):
    """2 ranks x 32 envs == 1 process x the same 64 envs: parameters after two
    iterations (3 + 3 epochs each), observation statistics and the normalised
    segment advantages agree to fp32 summation noise (the sharded run sums
    per-rank partial gradients / moments in a different order).  TCE (C2
    shape) with the overlapped and the serial update, and the BBRL agent
    (BASELINE configs[3])."""
    import numpy as np
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + (os.getpid() % 200) + (37 if overlap else 0) + \
        {"tce": 0, "tce_accrew": 53, "bbrl": 71, "bbrl_box_push": 83,
         "bbrl_table_tennis": 97}[kind]
    procs = [ctx.Process(target=_equiv_worker,
                         args=(r, 2, port, kind, overlap, q))
             for r in range(2)]
    procs.append(ctx.Process(target=_equiv_worker,
                             args=(0, 1, port, kind, overlap, q)))
    for p in procs:
        p.start()
    out = {(w, r): rest for w, r, *rest in
           (q.get(timeout=600) for _ in range(3))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (w0, a0, g0, b0, k0), (w1, a1, g1, b1, k1) = out[(2, 0)], out[(2, 1)]
    ws, as_, gs, bs, ks = out[(1, 0)]
    assert np.array_equal(w0, w1)                        # lock-step
    # the gradients went through the library's one-shot exchange (two processes,
    # buffers mapped through HIP IPC), not through torch.distributed
    assert k0 == k1 == "xgmi-oneshot" and ks == "none"
    # the balance check of iteration 1: reported by the sharded run too (ADVICE
    # r4), as the norms of the rank-averaged split gradients == the single
    # process's on the same 64 envs
    # (BBRL's first iteration: new == old, the trust region gradient is 0 and
    # the ratio inf on both sides)
    assert np.isfinite(bs[:2]).all() and bs[0] > 0 and b0 == b1
    np.testing.assert_allclose(b0[:2], bs[:2], rtol=2e-3, atol=1e-9)
    assert b0[2] == bs[2] or abs(b0[2] - bs[2]) <= 5e-3 * abs(bs[2])
    assert g0 == g1 == gs == 2 * N_GLOBAL * 500
    # normalised advantages: global-batch mean / std on both sides
    np.testing.assert_allclose(np.concatenate([a0, a1]), as_, rtol=2e-4,
                               atol=2e-4)
    # parameters (Adam normalises the gradient: an entry whose gradient is
    # pure summation noise moves by +-lr either way, hence the absolute term
    # 2 iterations x 3 epochs x lr 3e-4 would allow; observed: ~1e-6)
    np.testing.assert_allclose(w0, ws, rtol=2e-3, atol=2e-5)


def _rccl_one_rank_worker(port, q):
    """A ONE-rank RCCL world with TCE_FORCE_DIST=1: the sharded code path --
    both communicators, every all-reduce / all-gather / broadcast -- through
    the nccl backend on the one GPU of the box."""
    sys.path.insert(0, REPO)
    os.environ["TCE_FORCE_DIST"] = "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import datetime
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port,
                            rank=0, world_size=1,
                            timeout=datetime.timedelta(seconds=120),
                            device_id=torch.device("cuda", 0))
    from tce_rl_amd import dist as tdist
    from tce_rl_amd.config import bbrl_config, tce_config
    from tce_rl_amd.mp_exp import MPExperiment
    out = {}
    for name, cfg in (("tce", tce_config("metaworld", num_env=64, num_basis=5,
                                         epochs=3, evaluation_interval=0)),
                      ("bbrl", bbrl_config(num_env=64, epochs=3))):
        torch.manual_seed(3)
        exp = MPExperiment()
        exp.initialize(cfg, 0, None)
        assert exp.agent.dist.active and exp.agent.dist.world == 1
        exp.agent.step()
        tdist.reset_stats()
        res = exp.agent.step()
        flat = torch.cat([p.detach().reshape(-1).cpu() for p in
                          exp.agent.policy.parameters +
                          exp.agent.critic.parameters])
        kind = exp.agent.dist.exchange_kind()
        out[name] = (tdist.stats(), flat.numpy(),
                     float(res["critic_loss_mean"]), dist.get_backend(), kind)
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def _plain_worker(q):
    sys.path.insert(0, REPO)
    torch.cuda.set_device(0)
    from tce_rl_amd.config import bbrl_config, tce_config
    from tce_rl_amd.mp_exp import MPExperiment
    out = {}
    for name, cfg in (("tce", tce_config("metaworld", num_env=64, num_basis=5,
                                         epochs=3, evaluation_interval=0)),
                      ("bbrl", bbrl_config(num_env=64, epochs=3))):
        torch.manual_seed(3)
        exp = MPExperiment()
        exp.initialize(cfg, 0, None)
        exp.agent.step()
        exp.agent.step()
        flat = torch.cat([p.detach().reshape(-1).cpu() for p in
                          exp.agent.policy.parameters +
                          exp.agent.critic.parameters])
        out[name] = flat.numpy()
    q.put(out)


def test_sharded_path_through_a_one_rank_rccl_world():
    """the nccl-backend code path (communicator creation,
    the second communicator of the policy stream, gradient all-reduces,
    statistics gathers, pair broadcast) runs in the GPU test tier, and ends
    where the un-sharded process ends (one rank: the collectives are
    identities; the sharded path only un-fuses Adam)."""
    import numpy as np
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29850 + (os.getpid() % 100)
    p = ctx.Process(target=_rccl_one_rank_worker, args=(port, q))
    p.start()
    got = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    p = ctx.Process(target=_plain_worker, args=(q,))
    p.start()
    ref = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    for name in ("tce", "bbrl"):
        stats, flat, loss, backend, kind = got[name]
        assert backend == "nccl"
        # the gradients go through the library's own exchange (csrc/xchg.h),
        # counted like the torch.distributed collectives
        assert kind == "xgmi-oneshot"
        # 3 + 3 gradient all-reduces per step at least, plus statistics
        assert stats["collectives"] >= 6 and stats["bytes"] > 6 * 4 * 1000
        assert np.isfinite(flat).all() and np.isfinite(loss)
        np.testing.assert_allclose(flat, ref[name], rtol=2e-4, atol=2e-6)
