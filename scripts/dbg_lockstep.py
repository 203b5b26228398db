"""Debug: two ranks on one GPU, overlapped update; where do the replicas part?"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, overlap, steps, q):
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from tce_rl_amd.config import tce_config
    from tce_rl_amd.mp_exp import MPExperiment
    torch.manual_seed(100 + rank)
    cfg = tce_config("metaworld", num_env=64, num_basis=5, epochs=3, evaluation_interval=0, seed=0)
    cfg["params"]["agent"]["args"]["overlap_updates"] = overlap
    if os.environ.get("DBG_NOBAL"):
        cfg["params"]["agent"]["args"]["balance_check"] = None
    exp = MPExperiment()
    exp.initialize(cfg, 0, None)
    agent = exp.agent
    out = []
    for s in range(steps):
        agent.step()
        torch.cuda.synchronize()
        print("rank", rank, "step", s, "critic xchg", agent.xchg_critic.counters(), agent.xchg_critic.status(),
              "policy xchg", agent.xchg_policy.counters(), agent.xchg_policy.status(), flush=True)
        try:
            agent.flush_metrics()
        except RuntimeError as e:
            print("rank", rank, "ERR", str(e)[:80], flush=True)
        pol = torch.cat([p.detach().reshape(-1).cpu() for p in agent.policy.parameters])
        cri = torch.cat([p.detach().reshape(-1).cpu() for p in agent.critic.parameters])
        out.append((pol.numpy(), cri.numpy(), agent.policy_optimizer.flat_grad.cpu().numpy(),
                    agent.critic_optimizer.flat_grad.cpu().numpy()))
    st = (agent.xchg_critic.status() if agent.xchg_critic else -1, agent.xchg_policy.status() if agent.xchg_policy else -1)
    q.put((rank, out, st, agent.dist.exchange_kind()))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import numpy as np
    import torch.multiprocessing as mp
    overlap = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, 29711, bool(overlap), steps, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    (_, o0, st0, k0), (_, o1, st1, k1) = res
    print("status", st0, st1, k0, k1)
    for s, (a, b) in enumerate(zip(o0, o1)):
        for name, x, y in zip(("policy", "critic", "pgrad", "cgrad"), a, b):
            d = np.abs(x - y)
            nz = np.nonzero(d)[0]
            print("step", s, name, "max diff %.3e" % d.max(), "n differing", len(nz), "of", len(d),
                  "first idx", nz[:5].tolist(), "last idx", nz[-3:].tolist())
