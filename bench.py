#!/usr/bin/env python3
"""Benchmark of the TCE rollout + update hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step = one full ``TemporalCorrelatedAgent.step()`` without evaluation
(rollout on the synthetic env, obs RMS, critic forward, GAE + segment
advantage, 50 critic epochs, 50 trust-region-projected policy epochs) on the
BASELINE.json config 2 shape: 4096 envs per GPU, T = 500, P = 24, dof 4,
ProDMP with 5 basis functions (K = 24), fp32 (the reference accepts only
fp32/fp64).  N > 1: one process per GPU (torch.distributed.run), envs sharded
4096 per rank (weak scaling), one flat RCCL all-reduce of the gradients per
optimizer step.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

NUM_ENV, NUM_BASIS, EPOCHS = 4096, 5, 50
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_MFMA_PEAK_TF = 157.3       # MI355X_MICROARCH.md: FP32 matrix (dense, = vector peak)
F16_MFMA_PEAK_TF = 2500.0      # MI355X_MICROARCH.md: BF16/FP16 matrix, dense (no sparsity)


def build_agent(num_env, seed):
    from tce_rl_amd.config import tce_config
    from tce_rl_amd.mp_exp import MPExperiment
    cfg = tce_config("metaworld", num_env=num_env, num_basis=NUM_BASIS,
                     epochs=EPOCHS, dtype="float32", device="cuda", seed=seed,
                     evaluation_interval=0)
    exp = MPExperiment()
    exp.initialize(cfg, 0, None)
    return exp.agent, cfg


def kernel_time_us(fn, launches=20):
    """Average device time of `fn`'s kernel(s) with HIP events on the stream
    they are launched on.  The launches are queued behind a busy-wait kernel so
    that they run back to back (a Python launch loop on an idle GPU would time
    the host, not the kernel)."""
    fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    best = float("inf")
    for _ in range(3):
        torch.cuda._sleep(4_000_000)
        e0.record()
        for _ in range(launches):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / launches)
    return best


def kernel_time_cold_us(fn, launches=5):
    """Device time of ONE launch of `fn` right after a 1 GiB fill has displaced
    its inputs from the L2 / 256 MB Infinity Cache (per-launch HIP events)."""
    flush = torch.empty(1 << 28, device="cuda")
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    ts = []
    fn()
    for _ in range(launches):
        flush.fill_(1.0)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes
    (profiles/r01e_pmc.json, else r01c: rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE, FETCH x2 correction); None if absent."""
    for tag in ("r01e", "r01c"):
        try:
            with open(os.path.join(REPO, "profiles", tag + "_pmc.json")) as f:
                return json.load(f)["kernels"][kernel]["traffic_bytes"]
        except (OSError, KeyError, ValueError):
            continue
    return None


def roofline(agent):
    """Rooflines measured live with HIP events on the launch stream.

    dominant kernel = mlp_critic_bwd_kernel (the 50 critic epochs are ~90 % of
    the device time of a step): MFMA-bound, algorithmic flops per launch =
    6 * (D_in*H + H*H + H) per row (forward 2x, backward 4x) * N*T rows
    against the dense FP32 matrix peak.  GAE scan and trajectory generator:
    HBM-bound, algorithmic bytes per SURVEY 8(d)."""
    from tce_rl_amd import ops, critic_ops
    N, T = NUM_ENV, agent.sampler.num_times
    g = torch.Generator(device="cuda").manual_seed(0)
    net = agent.critic.net
    din = net.dim_in
    full = torch.randn(N, T + 1, din + 8, device="cuda", generator=g)
    xs = full[:, :-1, :din]
    rets = torch.randn(N, T, device="cuda", generator=g)
    saved = [p.grad for p in net.parameters()]
    run = critic_ops.EpochRunner(net)
    us_c = kernel_time_us(lambda: run.epoch(xs, rets, rets, 0.0), launches=5)
    run16 = critic_ops.EpochRunner(net, arith="f16x2")
    us_c16 = kernel_time_us(lambda: run16.epoch(xs, rets, rets, 0.0), launches=5)
    for p, gr in zip(net.parameters(), saved):
        p.grad = gr
    flops = 6.0 * (din * 128 + 128 * 128 + 128) * N * T
    critic = {"kernel": "mlp_critic_bwd_kernel<relu,10> (+ mlp_finish_kernel)",
              "bound": "mfma", "achieved": round(flops / us_c / 1e6, 2),
              "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s",
              "frac": round(flops / us_c / 1e6 / F32_MFMA_PEAK_TF, 4),
              "traffic": pmc_traffic("mlp_critic_bwd_kernel"),
              "us_per_launch": round(us_c, 1),
              "algorithmic_flops": flops, "dtype": "f32 (v_mfma_f32_16x16x4_f32)"}
    critic16 = {
        "kernel": "mlp_critic_bwd16_kernel<relu,2> (+ mlp_finish_kernel)",
        "bound": "mfma", "achieved": round(flops / us_c16 / 1e6, 2),
        "peak": F16_MFMA_PEAK_TF, "unit": "TFLOP/s",
        "frac": round(flops / us_c16 / 1e6 / F16_MFMA_PEAK_TF, 4),
        "traffic": pmc_traffic("mlp_critic_bwd16_kernel"),
        "us_per_launch": round(us_c16, 1),
        "algorithmic_flops": flops,
        "mfma_flops_issued": 3 * flops,
        "frac_issued": round(3 * flops / us_c16 / 1e6 / F16_MFMA_PEAK_TF, 4),
        "dtype": "f16x2 split operands, fp32 accumulate "
                 "(v_mfma_f32_16x16x32_f16; 3 MFMAs per product)"}
    del full, xs
    r = torch.randn(N, T, device="cuda", generator=g)
    v = torch.randn(N, T + 1, device="cuda", generator=g)
    d = torch.zeros(N, T, dtype=torch.bool, device="cuda")
    d[:, -1] = True
    tl = torch.zeros_like(d)
    us = kernel_time_us(lambda: ops.gae(r, v, d, tl, 1.0, 0.95, True))
    us_cold = kernel_time_cold_us(lambda: ops.gae(r, v, d, tl, 1.0, 0.95, True))
    alg = N * T * 18 + N * 4
    gae = {"kernel": "gae_dpp_kernel<float,true,true,8>", "bound": "hbm",
           "achieved": round(alg / us / 1e3, 1), "peak": HBM_PEAK_GBS,
           "unit": "GB/s", "frac": round(alg / us / 1e3 / HBM_PEAK_GBS, 4),
           "traffic": pmc_traffic("gae_dpp_kernel"),
           "us_per_launch": round(us, 2), "algorithmic_bytes": alg,
           "cold": {"us_per_launch": round(us_cold, 2),
                    "achieved": round(alg / us_cold / 1e3, 1),
                    "frac": round(alg / us_cold / 1e3 / HBM_PEAK_GBS, 4),
                    "note": "inputs evicted by a 1 GiB fill before the launch; "
                            "the back-to-back figure above re-reads its 37 MB "
                            "from the 256 MB Infinity Cache, as the step does "
                            "(values and rewards were just produced)"}}
    # trajectory generator (write-bound): T*2*dof*4 B written per env
    mp = agent.policy.mp
    K = mp.num_dof * mp.num_basis_g
    t0 = torch.zeros(N, device="cuda")
    times = ops.times(t0, mp.dt, T)
    w = 0.1 * torch.randn(N, K, device="cuda", generator=g)
    y0 = torch.rand(N, mp.num_dof, device="cuda", generator=g)
    v0 = torch.zeros(N, mp.num_dof, device="cuda")
    us2 = kernel_time_us(lambda: ops.prodmp_traj(mp, times, w, t0, y0, v0))
    us2_cold = kernel_time_cold_us(
        lambda: ops.prodmp_traj(mp, times, w, t0, y0, v0))
    alg2 = N * (T * 2 * mp.num_dof * 4 + 4 * (K + 2 * mp.num_dof + 1))
    extra = {"prodmp_traj": {
        "bound": "hbm", "achieved": round(alg2 / us2 / 1e3, 1),
        "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(alg2 / us2 / 1e3 / HBM_PEAK_GBS, 4),
        "traffic": pmc_traffic("prodmp_traj_kernel"),
        "us_per_launch": round(us2, 2), "algorithmic_bytes": alg2,
        "cold_us_per_launch": round(us2_cold, 2),
        "note": "trajectory kernel; the [T, 4+2(nb+1)] basis table "
                "(one 10 us kernel) is built once per time grid and reused by "
                "the ~100 trajectory / log-prob evaluations of a rollout + "
                "update (ops._times_flags)"}}
    extra["gae_scan"] = gae
    extra["critic_split_f16"] = critic16
    return critic, extra


def cpu_baseline():
    """The CPU oracle (torch-CPU restatement of the reference path, kind
    'port') on a bounded sample: 256 envs, full 50 + 50 epochs."""
    from tce_rl_amd.config import tce_config
    from oracle.agent_oracle import OracleTCE      # checker / baseline only
    n = 256
    # host cores this process may use (the GPU box gives 16 per GPU); torch
    # with more threads than cores thrashes on the small ops
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 16))
    torch.set_num_threads(threads)
    print("[bench] cpu_baseline: %d threads (%d visible cores)" %
          (threads, avail), file=sys.stderr, flush=True)
    cfg = tce_config("metaworld", num_env=n, num_basis=NUM_BASIS,
                     epochs=EPOCHS, device="cpu")
    o = OracleTCE(cfg["params"], n)
    t = time.perf_counter()
    steps = o.step()
    dt = time.perf_counter() - t
    return {"value": round(steps / dt, 1), "unit": "env-steps/s",
            "cores": threads, "kind": "port",
            "sample": "1 agent.step() of the torch-CPU oracle at %d envs "
                      "(T 500, 50 critic + 50 policy epochs), %.1f s" % (n, dt)}


def self_launch(args):
    """``python bench.py --gpus N`` from a plain shell (WORLD_SIZE unset): this
    process has not touched the GPU and never will -- it starts the N ranks as
    fresh children through torch.distributed.run (no exec), relays rank 0's
    JSON line and returns the children's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__),
           "--gpus", str(args.gpus), "--steps", str(args.steps),
           "--warmup", str(args.warmup)]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    if args.no_split_f16:
        cmd.append("--no-split-f16")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    line = None
    for ln in r.stdout.decode(errors="replace").splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    elif r.returncode == 0:
        print("[bench] no JSON line from rank 0", file=sys.stderr)
        return 1
    return r.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-split-f16", action="store_true",
                    help="skip the second timed region (critic_arith=f16x2)")
    args = ap.parse_args()
    # N > 1 without a launcher: become the launcher BEFORE any GPU call
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    # stdout carries exactly ONE line (the JSON record): everything else that
    # libraries print there (RCCL / gloo banners at communicator creation) is
    # sent to stderr by pointing fd 1 at fd 2 until the record is written
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" %
                         (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # TCE_BENCH_BACKEND=gloo: rehearsal of the N > 1 path with all ranks on the
    # GPUs that exist (one-GPU test box); the real run is nccl, one GPU per rank
    backend = os.environ.get("TCE_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    # TCE_FORCE_DIST=1: a world of ONE rank still creates the RCCL
    # communicators and takes the sharded code path (all-reduces included) --
    # the way to run the N > 1 path through RCCL on a one-GPU box
    force = world == 1 and os.environ.get("TCE_FORCE_DIST") == "1"
    if force:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ["RANK"], os.environ["WORLD_SIZE"] = "0", "1"
    if world > 1 or force:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(
                "cuda", local_rank))
        else:
            dist.init_process_group(backend)

    # the GLOBAL env count: MPExperiment gives every rank NUM_ENV of them and
    # the env / noise seed `seed + rank`
    agent, cfg = build_agent(NUM_ENV * world, seed=0)
    assert agent.sampler.num_env_train == NUM_ENV
    T = agent.sampler.num_times

    is_dist = dist.is_initialized()

    def barrier():
        if is_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def over_ranks(vals):
        """max over ranks of each value, plus every rank's first value."""
        if not is_dist:
            return vals, [vals[0]]
        tt = torch.tensor(vals, device="cuda", dtype=torch.float64)
        every = torch.empty(world, len(vals), device="cuda",
                            dtype=torch.float64)
        dist.all_gather_into_tensor(every, tt[None])
        return every.max(0).values.tolist(), every[:, 0].tolist()

    for _ in range(args.warmup):
        agent.step()
    barrier()
    if rank == 0:
        print("[bench] warmup done", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    pol_time = 0.0
    for _ in range(args.steps):
        res = agent.step()
        pol_time += res["update_policy_time"]
    barrier()
    elapsed = time.perf_counter() - t0
    (elapsed, pol_time), per_rank = over_ranks([elapsed, pol_time])

    # the same K steps (after W warm-up steps) with the critic epochs on the
    # split-f16 kernel (agent option critic_arith="f16x2"), reported beside
    # the fp32 figure as "split_f16_critic"
    fast = None
    if not args.no_split_f16:
        agent.critic_arith = "f16x2"
        agent._critic_split = 0
        for _ in range(args.warmup):
            agent.step()
        barrier()
        t1 = time.perf_counter()
        pol16 = 0.0
        for _ in range(args.steps):
            res = agent.step()
            pol16 += res["update_policy_time"]
        barrier()
        el16 = time.perf_counter() - t1
        fast, _ = over_ranks([el16, pol16])
        agent.critic_arith = "f32"

    if rank == 0:
        env_steps = world * NUM_ENV * T * args.steps
        print("[bench] timed region: %.3f s" % elapsed, file=sys.stderr,
              flush=True)
        roof, extra = roofline(agent)
        print("[bench] roofline done", file=sys.stderr, flush=True)
        out = {
            "metric": "env-steps/sec (TCE rollout + update, Metaworld-reach-"
                      "like, 4096 envs/GPU)",
            "value": round(env_steps / elapsed, 1),
            "unit": "env-steps/s",
            "policy_updates_per_sec": round(EPOCHS * args.steps / pol_time, 2),
            "iterations_per_sec": round(args.steps / elapsed, 4),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: TCE, Metaworld-"
                       "reach-like synthetic env, 4096 envs per GPU, T 500, "
                       "P 24, dof 4, ProDMP 5 basis (K 24), 50 critic + 50 "
                       "policy epochs, KL projection, fp32",
                       "num_env_per_gpu": NUM_ENV, "num_times": T,
                       "num_basis": NUM_BASIS, "epochs": EPOCHS,
                       "parallelism": "env-shard x%d" % world},
            "backend": (dist.get_backend() if is_dist else None),
            "rccl_ranks": (dist.get_world_size() if is_dist and
                           dist.get_backend() == "nccl" else 0),
            "ms_per_step_per_rank": [round(t / args.steps * 1e3, 2)
                                     for t in per_rank],
            "roofline": roof, "roofline_extra": extra,
        }
        if fast is not None:
            out["split_f16_critic"] = {
                "what": "the same workload with the 50 critic epochs on the "
                        "split-f16 matrix-core kernel (agent option "
                        "critic_arith=f16x2: fp32 operands carried as two f16 "
                        "parts, fp32 accumulate; parity-tested to the fp32 "
                        "kernel's own tolerances)",
                "value": round(env_steps / fast[0], 1), "unit": "env-steps/s",
                "ms_per_step": round(fast[0] / args.steps * 1e3, 2),
                "policy_updates_per_sec": round(
                    EPOCHS * args.steps / fast[1], 2),
                "roofline": extra["critic_split_f16"]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if is_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
