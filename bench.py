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
F64_MFMA_PEAK_TF = 78.6        # FP64 matrix = FP64 vector rate (v_mfma_f64_16x16x4_f64: 64 cycles per SIMD)


def build_agent(num_env, seed):
    from tce_rl_amd.config import tce_config
    from tce_rl_amd.mp_exp import MPExperiment
    cfg = tce_config("metaworld", num_env=num_env, num_basis=NUM_BASIS,
                     epochs=EPOCHS, dtype="float32", device="cuda", seed=seed,
                     evaluation_interval=0)
    exp = MPExperiment()
    exp.initialize(cfg, 0, None)
    return exp.agent, cfg


def time_balance_iteration(agent, sync):
    """Wall time (ms, synchronised on both sides) of ONE iteration with the
    policy balance check (num_iterations % balance_check == 1: every policy
    epoch also reports the gradient norms of the surrogate and of the trust
    region loss alone, mprl/rl/agent/temporal_correlated_agent.py:447-522).
    The reference's YAMLs set balance_check 25, so one iteration in 25 is of
    this kind; the timed window of the default command (iterations 6 - 25)
    holds none.  The check is triggered for the next iteration by a temporary
    balance_check = num_iterations ((n + 1) % n == 1)."""
    keep = agent.balance_check
    out = []
    for _ in range(3):
        agent.balance_check = agent.num_iterations
        assert (agent.num_iterations + 1) % agent.balance_check == 1
        sync()
        t = time.perf_counter()
        res = agent.step()
        sync()
        out.append((time.perf_counter() - t) * 1e3)
        assert "balance_ratio" in res, "not a balance-check iteration"
        agent.balance_check = None
        # ordinary iterations in between: the critic split of a balance
        # iteration is taken from the previous one's device times, which a
        # sharded run adopts two steps later (rl/agent.py:_adopt_split)
        for _ in range(3):
            agent.step()
    agent.balance_check = keep
    sync()
    return min(out)


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


PMC_TAGS = ("r06", "r05", "r04", "r03", "r02", "r01e", "r01c")
# (r05 / r06: the timed window only -- scripts/profiles_r06.sh, rocpd_stats.py
# --between-markers 1 2 --by-grid; earlier rounds: the whole run)
STATS_CSV = ("r06_bench_kernel_stats_by_grid.csv",
             "r05_bench_kernel_stats_by_grid.csv", "r04_bench_kernel_stats.csv",
             "r03_bench_kernel_stats.csv", "r02_bench_kernel_stats.csv")


def pmc_lookup(kernel):
    """(HBM bytes per launch, source file) of `kernel` from the COMMITTED PMC
    passes (profiles/<tag>_pmc.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate passes, FETCH x2 correction on gfx950) -- read from the file, not
    measured in this run; (None, None) if absent."""
    for tag in PMC_TAGS:
        try:
            rel = os.path.join("profiles", tag + "_pmc.json")
            with open(os.path.join(REPO, rel)) as f:
                return json.load(f)["kernels"][kernel]["traffic_bytes"], rel
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def mfma_busy(kernel):
    """(matrix-pipe utilisation of `kernel` alone on the chip, source file) from
    the COMMITTED counter pass profiles/r06_pmc_mfma.json (else r04's)
    (SQ_VALU_MFMA_BUSY_CYCLES per SIMD / GRBM_GUI_ACTIVE per XCD) -- read from
    the file, not measured in this run; (None, None) if absent."""
    for tag in ("r06", "r04"):
        try:
            rel = os.path.join("profiles", tag + "_pmc_mfma.json")
            with open(os.path.join(REPO, rel)) as f:
                return json.load(f)["kernels"][kernel]["mfma_busy"], rel
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def pmc_traffic(kernel):
    return pmc_lookup(kernel)[0]


def mlpb_pmc(*keys):
    """A value of the committed counter summary of the three-part bf16 critic
    kernel (profiles/r05_pmc_mlpb.json: scripts/pmc_mlpb.sh), None if absent."""
    try:
        with open(os.path.join(REPO, "profiles", "r05_pmc_mlpb.json")) as f:
            v = json.load(f)
        for k in keys:
            v = v[k]
        return v
    except (OSError, KeyError, ValueError):
        return None


def pmc_source(kernel):
    return pmc_lookup(kernel)[1]


# (the critic's backward instance; "<1, 10, true>" is the policy's hidden-layer mode)
DOMINANT = "mlp_critic_bwd_kernel<1, 10, false>"


def profiled_us(kernel_substr):
    """(average duration in us, source file) of a kernel in the COMMITTED
    rocprofv3 --kernel-trace --stats summary of this same command
    (profiles/r03_bench_kernel_stats.csv, else r02) -- read from the file, not
    measured in this run; (None, None) if absent."""
    import csv
    for name in STATS_CSV:
        try:
            rel = os.path.join("profiles", name)
            with open(os.path.join(REPO, rel)) as f:
                # (a by-grid summary has one row per workgroup count: the
                # call-weighted average over them)
                tot = calls = 0.0
                for row in csv.DictReader(f):
                    if kernel_substr in row["Name"]:
                        tot += float(row["TotalDurationNs"])
                        calls += float(row["Calls"])
                if calls:
                    return round(tot / calls / 1e3, 1), rel
        except (OSError, KeyError, ValueError):
            pass
    return None, None


def wide_traffic(tag, rows):
    """(HBM bytes per epoch of the wide critic -- chain + gradient kernel --,
    source) from the committed PMC passes, which ran at 819 200 rows (C3);
    other row counts are scaled (the traffic is per-row activations)."""
    key = "<float" if tag in ("f32", "float32") else "<double"
    a, src = pmc_lookup("mlpw_chain_kernel" + key)
    b, _ = pmc_lookup("mlpw_grad_kernel" + key)
    if a is None or b is None:
        return None, None
    if rows != 819200:
        return int((a + b) * rows / 819200), src + " (819 200 rows, scaled to %d)" % rows
    return a + b, src


def hbm_entry(name, alg, us, us_cold=None, note=None, pmc_kernel=None):
    traffic, src = pmc_lookup(pmc_kernel) if pmc_kernel else (None, None)
    d = {"kernel": name, "bound": "hbm", "achieved": round(alg / us / 1e3, 1),
         "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(alg / us / 1e3 / HBM_PEAK_GBS, 4), "traffic": traffic,
         "traffic_source": src,
         "us_per_launch": round(us, 2), "algorithmic_bytes": alg}
    if us_cold is not None:
        d["cold"] = {"us_per_launch": round(us_cold, 2),
                     "achieved": round(alg / us_cold / 1e3, 1),
                     "frac": round(alg / us_cold / 1e3 / HBM_PEAK_GBS, 4)}
    if note:
        d["note"] = note
    return d


def roofline(agent, critic_ms_in_step, with_f16=False, envs_in_step=None):
    """Rooflines measured live with HIP events on the launch stream.

    dominant kernel = mlp_critic_bwd_kernel (+ mlp_finish_kernel: the 50 critic
    epochs are ~90 % of the device time of a step): MFMA-bound, algorithmic
    flops per launch = 6 * (D_in*H + H*H + H) per row (forward 2x, backward
    4x) * N*T rows against the dense FP32 matrix peak.  `achieved` uses the
    average launch duration INSIDE the timed steps (HIP events on the critic's
    stream around its 50 epochs, policy stream running beside it -- what
    rocprofv3 sees for the same command); the isolated back-to-back figure is
    reported beside it.  GAE scan, trajectory generator and env rollout:
    HBM-bound, algorithmic bytes per SURVEY 8(d), at the C2 working set (which
    stays in the 256 MB Infinity Cache between launches) AND at 8x the envs
    (295 / 529 MB per launch: past the cache)."""
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
    us_c16 = None
    if with_f16:
        run16 = critic_ops.EpochRunner(net, arith="f16x2")
        us_c16 = kernel_time_us(lambda: run16.epoch(xs, rets, rets, 0.0),
                                launches=5)
    runb = critic_ops.EpochRunner(net, arith="bf16x3")
    us_cb = kernel_time_us(lambda: runb.epoch(xs, rets, rets, 0.0), launches=5)
    for p, gr in zip(net.parameters(), saved):
        p.grad = gr
    flops = 6.0 * (din * 128 + 128 * 128 + 128) * N * T
    # what the kernel EXECUTES per row: the forward pass
    # 2 (din H + H H + H), dH1 = W2^T dY2 and dW2 (2 H H each), dW1 with db1 as
    # a column of ones (2 (din + 1) H), dw3 (2 H) -- no input gradient of layer
    # 1, which SURVEY 8d's "3 x forward" convention counts; mfma_busy (the
    # matrix pipe's busy cycles, counters) is to be held against frac_executed
    flops_exec = 2.0 * (din * 128 + 128 * 128 + 128 + 2 * 128 * 128 +
                        (din + 1) * 128 + 128) * N * T
    # the timed steps hold envs_in_step envs per rank (4096 unless --scaling strong)
    flops_step = flops * (envs_in_step or N) / N
    us_step = critic_ms_in_step * 1e3 / EPOCHS
    critic = {"kernel": "mlp_critic_bwd_kernel<relu,10> (+ mlp_finish_kernel)",
              "bound": "mfma", "achieved": round(flops_step / us_step / 1e6, 2),
              "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s",
              "frac": round(flops_step / us_step / 1e6 / F32_MFMA_PEAK_TF, 4),
              "traffic": pmc_traffic("mlp_critic_bwd_kernel"),
              "traffic_source": pmc_source("mlp_critic_bwd_kernel"),
              "us_per_launch": round(us_step, 1),
              "measured": "HIP events on the critic stream around the 50 "
                          "epochs of every timed step (launch + slab "
                          "reduction / Adam), policy epochs on a second stream",
              "isolated_back_to_back": {
                  "us_per_launch": round(us_c, 1),
                  "achieved": round(flops / us_c / 1e6, 2),
                  "frac": round(flops / us_c / 1e6 / F32_MFMA_PEAK_TF, 4)},
              "mfma_busy": mfma_busy("mlp_critic_bwd_kernel")[0],
              "mfma_busy_source": mfma_busy("mlp_critic_bwd_kernel")[1],
              "rocprof_us_per_launch": profiled_us(DOMINANT)[0],
              "rocprof_source": profiled_us(DOMINANT)[1],
              # the same algorithmic flops / the committed trace's average
              # duration of this kernel (timed window of the same command)
              "frac_from_profile": None if not profiled_us(DOMINANT)[0]
              else round(flops / profiled_us(DOMINANT)[0] / 1e6
                         / F32_MFMA_PEAK_TF, 4),
              "algorithmic_flops": flops_step,
              "executed_flops": flops_exec * (envs_in_step or N) / N,
              "frac_executed": round(flops_exec * (envs_in_step or N) / N
                                     / us_step / 1e6 / F32_MFMA_PEAK_TF, 4),
              "dtype": "f32 (v_mfma_f32_16x16x4_f32)"}
    if getattr(agent, "critic_arith", "f32") == "bf16x3":
        # --critic-arith bf16x3: the timed steps ran csrc/mlpb.hip; algorithmic
        # (fp32-equivalent) flops against the dense bf16 matrix peak, the six
        # issued MFMAs per product beside it
        a_ = flops_step / us_step / 1e6
        critic.update({
            "kernel": "mlp_critic_bwdb_kernel<relu,2> (+ mlp_finish_kernel)",
            "achieved": round(a_, 2), "peak": F16_MFMA_PEAK_TF,
            "frac": round(a_ / F16_MFMA_PEAK_TF, 4),
            "frac_issued": round(6 * a_ / F16_MFMA_PEAK_TF, 4),
            "of_fp32_mfma_peak": round(a_ / F32_MFMA_PEAK_TF, 3),
            "traffic": mlpb_pmc("hbm", "traffic_bytes"),
            "traffic_source": "profiles/r05_pmc_mlpb.json",
            "isolated_back_to_back": {
                "us_per_launch": round(us_cb, 1),
                "achieved": round(flops / us_cb / 1e6, 2),
                "frac": round(flops / us_cb / 1e6 / F16_MFMA_PEAK_TF, 4)},
            "mfma_busy": mlpb_pmc("mfma_busy"),
            "mfma_busy_source": "profiles/r05_pmc_mlpb.json",
            "rocprof_us_per_launch": None, "rocprof_source": None,
            "frac_from_profile": None,
            "dtype": "bf16x3 operands (24 bits, fp32 range), six partial "
                     "products, fp32 accumulate (v_mfma_f32_16x16x32_bf16)"})
    critic16 = None if us_c16 is None else {
        "kernel": "mlp_critic_bwd16_kernel<relu,2> (+ mlp_finish_kernel)",
        "bound": "mfma", "achieved": round(flops / us_c16 / 1e6, 2),
        "peak": F16_MFMA_PEAK_TF, "unit": "TFLOP/s",
        "frac": round(flops / us_c16 / 1e6 / F16_MFMA_PEAK_TF, 4),
        "traffic": pmc_traffic("mlp_critic_bwd16_kernel"),
        "traffic_source": pmc_source("mlp_critic_bwd16_kernel"),
        "us_per_launch": round(us_c16, 1),
        "algorithmic_flops": flops,
        "mfma_flops_issued": 3 * flops,
        "frac_issued": round(3 * flops / us_c16 / 1e6 / F16_MFMA_PEAK_TF, 4),
        "dtype": "f16x2 split operands, fp32 accumulate "
                 "(v_mfma_f32_16x16x32_f16; 3 MFMAs per product)"}
    del full, xs
    extra = {} if critic16 is None else {"critic_split_f16": critic16}
    extra["critic_bf16x3"] = {
        "kernel": "mlp_critic_bwdb_kernel<relu,2> (+ mlp_finish_kernel)",
        "bound": "mfma", "achieved": round(flops / us_cb / 1e6, 2),
        "peak": F16_MFMA_PEAK_TF, "unit": "TFLOP/s",
        "frac": round(flops / us_cb / 1e6 / F16_MFMA_PEAK_TF, 4),
        "traffic": mlpb_pmc("hbm", "traffic_bytes"),
        "traffic_source": "profiles/r05_pmc_mlpb.json",
        "mfma_busy": mlpb_pmc("mfma_busy"),
        "us_per_launch": round(us_cb, 1),
        "algorithmic_flops": flops,
        "mfma_flops_issued": 6 * flops,
        "frac_issued": round(6 * flops / us_cb / 1e6 / F16_MFMA_PEAK_TF, 4),
        "speedup_over_exact_fp32_kernel": round(us_c / us_cb, 3),
        "of_fp32_mfma_peak": round(flops / us_cb / 1e6 / F32_MFMA_PEAK_TF, 3),
        "dtype": "bf16x3: three-part bf16 operands (24 bits, fp32 range), six "
                 "partial products, fp32 accumulate (v_mfma_f32_16x16x32_bf16)",
        "note": "isolated back to back, C2 rows; agent option critic_arith="
                "bf16x3 (configs.C2_bf16x3_critic times whole steps with it); "
                "bound by the VALU work of the three-way splits and the LDS "
                "reads that share the issue port with the MFMAs at one wave per "
                "SIMD (KERNELS.md), not by the matrix cores"}

    def gae_case(n):
        r = torch.randn(n, T, device="cuda", generator=g)
        v = torch.randn(n, T + 1, device="cuda", generator=g)
        d = torch.zeros(n, T, dtype=torch.bool, device="cuda")
        d[:, -1] = True
        tl = torch.zeros_like(d)
        f = lambda: ops.gae(r, v, d, tl, 1.0, 0.95, True)
        return n * T * 18 + n * 4, kernel_time_us(f), kernel_time_cold_us(f)
    alg, us, us_cold = gae_case(N)
    extra["gae_scan"] = hbm_entry(
        "gae_dpp_kernel<float,true,true,8>", alg, us, us_cold,
        "C2 size: the 37 MB working set stays in the 256 MB Infinity Cache "
        "between back-to-back launches (as in the step, where values and "
        "rewards were just produced); `cold` = after a 1 GiB fill",
        "gae_dpp_kernel")
    alg, us, us_cold = gae_case(8 * N)
    extra["gae_scan_32768_envs"] = hbm_entry(
        "gae_dpp_kernel<float,true,true,8>", alg, us, us_cold,
        "295 MB per launch: past the Infinity Cache")

    # trajectory generator (write-bound): T*2*dof*4 B written per env
    def traj_case(mp, n, t_len):
        K = mp.num_dof * mp.num_basis_g
        t0 = torch.zeros(n, device="cuda")
        times = ops.times(t0, mp.dt, t_len)
        w = 0.1 * torch.randn(n, K, device="cuda", generator=g)
        y0 = torch.rand(n, mp.num_dof, device="cuda", generator=g)
        v0 = torch.zeros(n, mp.num_dof, device="cuda")
        f = lambda: ops.prodmp_traj(mp, times, w, t0, y0, v0)
        alg = n * (t_len * 2 * mp.num_dof * 4 + 4 * (K + 2 * mp.num_dof + 1))
        return alg, kernel_time_us(f), kernel_time_cold_us(f)
    mp = agent.policy.mp
    alg, us, us_cold = traj_case(mp, N, T)
    extra["prodmp_traj"] = hbm_entry(
        "prodmp_traj_rows_kernel<float,4,6>", alg, us, us_cold,
        "trajectory kernel; the [T, 4+2(nb+1)] basis table (one 10 us kernel) "
        "is built once per time grid and reused by the ~100 trajectory / "
        "log-prob evaluations of a rollout + update (ops._times_flags)",
        "prodmp_traj_rows_kernel<float, 4")
    alg, us, us_cold = traj_case(mp, 8 * N, T)
    extra["prodmp_traj_32768_envs"] = hbm_entry(
        "prodmp_traj_rows_kernel<float,4,6>", alg, us, us_cold,
        "529 MB per launch: past the Infinity Cache")
    from tce_rl_amd.mp import ProDMP
    mp7 = ProDMP(num_dof=7, num_basis=8, tau=2.0, alpha_phase=3, alpha=10,
                 dt=0.02, basis_bandwidth_factor=3, weights_scale=0.3,
                 goal_scale=0.3, dtype=torch.float32, device="cuda")
    alg, us, us_cold = traj_case(mp7, 16 * N, 100)
    extra["prodmp_traj_dof7"] = hbm_entry(
        "prodmp_traj_rows_kernel<float,7,9>", alg, us, us_cold,
        "BASELINE configs[2] rows (dof 7: 56-byte rows, stored through a "
        "wave-private LDS slab as contiguous 8-byte chunks), 65536 envs x T "
        "100 = 367 MB", "prodmp_traj_rows_kernel<float, 7")

    # env rollout kernel: writes the [N, T+1, D] state buffer once, reads the
    # desired trajectory; column moments in the same pass
    env = agent.sampler.train_envs
    D = env.dim_obs
    acts = torch.randn(N, T, 2 * env.num_dof, device="cuda", generator=g)
    obs0 = env.reset()
    shift = torch.zeros(D, device="cuda")
    f = lambda: ops.env_rollout(acts, obs0, env.task, env.num_dof,
                                env.dim_task_obs, env.dt, 400.0, 40.0,
                                want_states=True, shift=shift,
                                want_moments=True)
    alg = N * ((T + 1) * D * 4 + T * 2 * env.num_dof * 4 + T * 4)
    extra["env_rollout"] = hbm_entry(
        "env_rollout_kernel<float>", alg, kernel_time_us(f, launches=5),
        kernel_time_cold_us(f),
        "one launch = one whole episode of the %d synthetic envs: state "
        "buffer written once (%d MB), observation moments accumulated in the "
        "same pass; instruction-bound (one wave per env, ~200 instructions "
        "per step)" % (N, N * (T + 1) * D * 4 // 1000000),
        "env_rollout_kernel")
    del acts

    # wide / fp64 critics of BASELINE configs[2] (box pushing): one epoch =
    # chain kernel + weight-gradient kernel (+ slab reduction)
    from tce_rl_amd.nn import MLP
    for tag, dtype, peak in (("f32", torch.float32, F32_MFMA_PEAK_TF),
                             ("f64", torch.float64, F64_MFMA_PEAK_TF)):
        wide = MLP("ValueFunction", 22, 1, [256, 256], "orthogonal", 1.0,
                   "leaky_relu", None, dtype, torch.device("cuda"))
        n3, t3 = 8192, 100
        st = torch.randn(n3, t3 + 1, 36, device="cuda", generator=g,
                         dtype=dtype)[:, :-1, :22]
        rt = torch.randn(n3, t3, device="cuda", generator=g, dtype=dtype)
        runw = critic_ops.make_runner(wide)
        # (12 launches: three 3 ms launches right after the busy-wait kernel run
        # before the clocks are back up -- 10 % slower than steady state)
        us_w = kernel_time_us(lambda: runw.epoch(st, rt, rt, 0.0), launches=12)
        fl = 6.0 * (22 * 256 + 256 * 256 + 256) * n3 * t3
        extra["critic_256x2_" + tag] = {
            "kernel": "mlpw_chain_kernel + mlpw_grad_kernel + "
                      "mlpw_finish_kernel (<%s, 256>)" % tag,
            "bound": "mfma", "achieved": round(fl / us_w / 1e6, 2),
            "peak": peak, "unit": "TFLOP/s",
            "frac": round(fl / us_w / 1e6 / peak, 4),
            "traffic": wide_traffic(tag, n3 * t3)[0],
            "traffic_source": wide_traffic(tag, n3 * t3)[1],
            "us_per_epoch": round(us_w, 1), "algorithmic_flops": fl,
            "mfma_busy": {"chain": mfma_busy("mlpw_chain_kernel<" + (
                "float" if tag == "f32" else "double"))[0],
                "grad": mfma_busy("mlpw_grad_kernel<" + (
                    "float" if tag == "f32" else "double"))[0],
                "source": mfma_busy("mlpw_chain_kernel<float")[1]},
            "workload": "BASELINE configs[2] critic: 8192 envs x T 100 rows, "
                        "D_in 22 -> 256 -> 256 -> 1, leaky_relu",
            "dtype": tag + (" (v_mfma_f32_16x16x4_f32)" if tag == "f32"
                            else " (v_mfma_f64_16x16x4_f64)")}
        del st, rt, runw, wide
    return critic, extra


# The other BASELINE.json configs, a few timed steps each.
# `kind`: "tce" -> tce_config(env, ...), "bbrl" -> bbrl_config(...).  C4 / C5
# are multi-GPU configs in BASELINE.json (4 x 4096 / 8 x 4096 envs): what one
# GPU runs of them -- its 4096-env shard -- is timed here.
OTHER_CONFIGS = [
    ("C3_box_push_f32", dict(
        kind="tce", env="box_push", num_env=8192, num_basis=8, epochs=50,
        dtype="float32",
        workload="BASELINE.json configs[2]: TCE, box-pushing-like synthetic "
                 "env, 8192 envs, T 100, dof 7, ProDMP 8 basis (K 63), critic "
                 "22 -> 256 -> 256 -> 1 leaky_relu, 50 + 50 epochs, fp32 "
                 "(mprl/config/box_push_random_init/tcp/entire/shared.yaml)")),
    ("C3_box_push_f64", dict(
        kind="tce", env="box_push", num_env=8192, num_basis=8, epochs=50,
        dtype="float64",
        workload="the same in float64, the reference's dtype for this task "
                 "(box_push_random_init/tcp/entire/shared.yaml:7)")),
    ("C4_bbrl_shard", dict(
        kind="bbrl", num_env=4096, epochs=100, dtype="float32",
        workload="BASELINE.json configs[3]: BBRL, Metaworld-push-like "
                 "synthetic black-box env, one GPU's 4096-env shard of the "
                 "16384, K 20 diagonal, 32 x 2 nets, 100 + 100 epochs "
                 "(mprl/config/metaworld/bbrl/entire/shared.yaml:38-39), "
                 "trust_region_coeff 10, set_variance, fp32")),
    ("C5_table_tennis_nb3_shard", dict(
        kind="tce", env="table_tennis", num_env=4096, num_basis=3, epochs=50,
        dtype="float32",
        workload="BASELINE.json configs[4] with the reference's 3 basis "
                 "functions (K 28; table_tennis_4d/tcp/entire/shared.yaml:"
                 "56-60): TCE, table-tennis-like synthetic env, one GPU's "
                 "4096-env shard of the 32768, T 350, dof 7, MDP-reward, "
                 "tanh 256 x 1 policy, leaky_relu 256 x 2 critic, fp32")),
    ("C5_table_tennis_nb8_shard", dict(
        kind="tce", env="table_tennis", num_env=4096, num_basis=8, epochs=50,
        dtype="float32",
        workload="BASELINE.json configs[4] as worded (ProDMP 8 basis, K 63), "
                 "otherwise as above")),
    # NOT the headline's instruction: the same C2 steps with the 50 critic epochs on
    # the bf16 matrix cores with THREE-PART operands (csrc/mlpb.hip): every fp32
    # operand as b0 + b1 + b2 exactly (24 bits, fp32's exponent range), six partial
    # products per product, fp32 accumulate -- not narrower than fp32 and as close to
    # the fp64 truth as the exact-fp32 kernel (tests/test_mlpb_gpu.py), but a change
    # of arithmetic all the same: an extra entry, `value` stays on v_mfma_f32_16x16x4_f32
    ("C2_bf16x3_critic", dict(
        kind="tce", env="metaworld", num_env=4096, num_basis=5, epochs=50,
        dtype="float32", critic_arith="bf16x3",
        workload="configs[1] (TCE Metaworld-reach-like, 4096 envs, T 500, K 24, "
                 "50 + 50 epochs) with agent option critic_arith=bf16x3: every "
                 "fp32 operand of the critic epoch carried as three bf16 parts "
                 "(x = b0 + b1 + b2 exactly: 24 bits, fp32's range), six of the "
                 "nine partial products (the dropped ones < 2^-25 of the "
                 "product), fp32 accumulate, on v_mfma_f32_16x16x32_bf16 "
                 "(csrc/mlpb.hip); loss and gradients no further from an fp64 "
                 "reference than the exact-fp32 kernel's "
                 "(tests/test_mlpb_gpu.py)")),
    # NOT the headline's arithmetic: the same C2 steps with the 50 critic epochs on
    # the split-f16 matrix-core kernel (22-bit operands, fp32 accumulate: narrower
    # than the reference's fp32, hence an option and an extra entry, never `value`)
    ("C2_split_f16_critic", dict(
        kind="tce", env="metaworld", num_env=4096, num_basis=5, epochs=50,
        dtype="float32", critic_arith="f16x2",
        workload="NOT BASELINE arithmetic -- configs[1] (TCE Metaworld-reach-"
                 "like, 4096 envs, T 500, K 24, 50 + 50 epochs) with agent "
                 "option critic_arith=f16x2: every fp32 operand of the critic "
                 "epoch carried as two f16 parts (22 bits), fp32 accumulate, "
                 "on v_mfma_f32_16x16x32_f16 (csrc/mlp16.hip); parity-tested to "
                 "the fp32 kernel's own error bounds, but narrower than fp32")),
    # the reference's CLASS DEFAULT num_minibatchs = 10 (every shipped YAML sets 1):
    # 50 epochs x 10 optimizer steps on gathered rows, one C call per epoch
    # (tce_mlp_critic_minibatch_f32).  Twice: with the reference's own permutation
    # draw (numpy's global MT19937 Fisher-Yates over 2 M rows on the host --
    # sequential by construction, ~28 ms per epoch: that entry is HOST-bound and
    # says so) and with a keyed Feistel permutation computed on the device
    ("C2_minibatch10", dict(
        kind="tce", env="metaworld", num_env=4096, num_basis=5, epochs=50,
        dtype="float32", num_minibatchs=10, steps=2, warmup=2,
        workload="configs[1] with the reference's class default num_minibatchs "
                 "= 10 (temporal_correlated_agent.py:25,343-366): 500 critic "
                 "steps per iteration on gathered 204 800-row pieces, the "
                 "permutations drawn as the reference draws them "
                 "(np.random.shuffle on the host: the step is bound by that "
                 "draw, see host_permutation_ms_per_step)")),
    ("C2_minibatch10_device_perm", dict(
        kind="tce", env="metaworld", num_env=4096, num_basis=5, epochs=50,
        dtype="float32", num_minibatchs=10, minibatch_permutation="device",
        workload="the same with agent option minibatch_permutation=device "
                 "(a keyed Feistel permutation computed on the GPU, "
                 "tce_feistel_permutation: not numpy's sequence)")),
    # the multi-GPU configs at their FULL size on this one GPU (they fit: 288 GB):
    # the denominators of the strong-scaling curves (N GPUs x N-th of the envs)
    ("C4_bbrl_full_16384", dict(
        kind="bbrl", num_env=16384, epochs=100, dtype="float32",
        workload="BASELINE.json configs[3] at its full size on ONE GPU: BBRL, "
                 "16384 envs, otherwise as C4_bbrl_shard")),
    ("C5_table_tennis_nb3_full_32768", dict(
        kind="tce", env="table_tennis", num_env=32768, num_basis=3, epochs=50,
        dtype="float32", steps=2, warmup=2,
        workload="BASELINE.json configs[4] at its full size on ONE GPU: TCE, "
                 "32768 envs x T 350 (11.5 M critic rows per epoch), "
                 "otherwise as C5_table_tennis_nb3_shard")),
]


def build_config_agent(spec, seed=0):
    from tce_rl_amd.config import bbrl_config, tce_config
    from tce_rl_amd.mp_exp import MPExperiment
    if spec["kind"] == "bbrl":
        cfg = bbrl_config(num_env=spec["num_env"], epochs=spec["epochs"],
                          dtype=spec["dtype"], device="cuda", seed=seed,
                          evaluation_interval=0)
    else:
        cfg = tce_config(spec["env"], num_env=spec["num_env"],
                         num_basis=spec["num_basis"], epochs=spec["epochs"],
                         dtype=spec["dtype"], device="cuda", seed=seed,
                         evaluation_interval=0)
        if spec.get("critic_arith"):
            cfg["params"]["agent"]["args"]["critic_arith"] = spec["critic_arith"]
    for k in ("num_minibatchs", "minibatch_permutation"):
        if k in spec:
            cfg["params"]["agent"]["args"][k] = spec[k]
    exp = MPExperiment()
    exp.initialize(cfg, 0, None)
    return exp.agent


def run_config(name, spec, steps, warmup):
    steps, warmup = spec.get("steps", steps), spec.get("warmup", warmup)
    return _run_config(name, spec, steps, warmup)


def _run_config(name, spec, steps, warmup):
    """One entry of the `configs` block: W untimed + K timed agent.step()s of
    one BASELINE config on this GPU, synchronised wall time, plus the roofline
    of its dominant kernel from the device time (HIP events on the critic's
    stream) of the critic epochs INSIDE those steps."""
    from tce_rl_amd import mlp_ops
    mlp_ops.LIBRARY_CALLS.clear()
    agent = build_config_agent(spec)
    T = getattr(agent.sampler, "num_times", None) or \
        agent.sampler.train_envs.num_times
    N = spec["num_env"]
    for _ in range(warmup):
        agent.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    crit = pol = samp = 0.0
    results = []
    for _ in range(steps):
        results.append(agent.step())       # metrics are read after the timed region
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    for res in results:
        crit += res.get("update_critic_time", 0.0)
        pol += res.get("update_policy_time", 0.0)
        samp += res.get("sampling_time", 0.0)
    E = spec["epochs"]
    bal_ms = None
    # (black-box agent: black_box_agent.py:218-284, two extra launches per
    # policy epoch without the optimizer step)
    if isinstance(agent.balance_check, int):
        bal_ms = time_balance_iteration(agent, torch.cuda.synchronize)
    out = {"workload": spec["workload"], "num_env": N, "num_times": T,
           "epochs": "%d + %d" % (E, E), "dtype": spec["dtype"],
           "steps": steps, "warmup": warmup,
           "ms_per_step": round(el / steps * 1e3, 2),
           "env_steps_per_sec": round(N * T * steps / el, 1),
           "sampling_ms": round(samp / steps * 1e3, 2)}
    if spec.get("num_minibatchs", 1) > 1:
        out["num_minibatchs"] = spec["num_minibatchs"]
        out["minibatch_permutation"] = spec.get("minibatch_permutation",
                                                "numpy")
        if out["minibatch_permutation"] == "numpy":
            import numpy as np
            rows_mb = N * T if spec["kind"] == "tce" else N
            tp = time.perf_counter()
            for _ in range(3):
                np.random.shuffle(np.arange(rows_mb))
            out["host_permutation_ms_per_step"] = round(
                (time.perf_counter() - tp) / 3 * 1e3 * spec["epochs"], 1)
    if bal_ms is not None:
        bc = agent.balance_check
        out["balance_check_iteration_ms"] = round(bal_ms, 2)
        out["ms_per_step_amortised"] = round(
            ((bc - 1) * out["ms_per_step"] + bal_ms) / bc, 2)
    net = agent.critic.net
    hs = [l.weight.shape[0] for l in net.layers[:-1]]
    din = net.dim_in
    rows = N * T if spec["kind"] == "tce" else N
    flops = 6.0 * (din * hs[0] + hs[0] * hs[1] + hs[1]) * rows
    f64 = spec["dtype"] == "float64"
    peak = F64_MFMA_PEAK_TF if f64 else F32_MFMA_PEAK_TF
    if spec.get("critic_arith") in ("f16x2", "bf16x3"):
        # algorithmic (fp32-equivalent) flops against the dense f16 / bf16 matrix
        # peak; the kernels issue three (f16x2) / six (bf16x3) MFMAs per product
        peak = F16_MFMA_PEAK_TF
    if spec["kind"] == "tce":
        us = crit / steps / E * 1e6
        out["policy_updates_per_sec"] = round(E * steps / pol, 2)
        out["critic_ms_per_step"] = round(crit / steps * 1e3, 2)
        out["policy_ms_per_step"] = round(pol / steps * 1e3, 2)
        out["roofline"] = {
            "kernel": "critic epoch %d -> %d -> %d -> 1 (%s), %d rows"
                      % (din, hs[0], hs[1], getattr(
                          agent, "_critic_runner", None).__class__.__name__,
                         rows),
            "bound": "mfma", "achieved": round(flops / us / 1e6, 2),
            "peak": peak, "unit": "TFLOP/s",
            "frac": round(flops / us / 1e6 / peak, 4),
            "us_per_launch": round(us, 1), "algorithmic_flops": flops,
            "traffic": wide_traffic(spec["dtype"], rows)[0]
            if hs[0] == 256 else None,
            "traffic_source": wide_traffic(spec["dtype"], rows)[1]
            if hs[0] == 256 else None,
            "mfma_busy": None if hs[0] != 256 else {
                "chain": mfma_busy("mlpw_chain_kernel<" + (
                    "double" if f64 else "float"))[0],
                "grad": mfma_busy("mlpw_grad_kernel<" + (
                    "double" if f64 else "float"))[0],
                "source": mfma_busy("mlpw_chain_kernel<float")[1]},
            "measured": "HIP events around the %d critic epochs of every "
                        "timed step, policy epochs on a second stream" % E}
    else:
        # the black-box agent's step: 2 x E epochs on 4096 rows of 32-wide nets
        # are latency-bound (56 MFLOP per epoch); the HBM work is the rollout
        upd = res.get("update_time", 0.0)
        out["update_ms_last_step"] = round(upd * 1e3, 2)
        out["policy_updates_per_sec"] = round(
            E / max(upd, 1e-9), 2) if upd else None
        # the floor of the dependent chain: the same update on ONE workgroup's
        # worth of rows (64 envs = one 64-row tile per row kernel): every kernel
        # of an epoch then is its own critical path -- 3 layers forward, the
        # head, 3 layers backward, the slab reduction, the finish -- plus the
        # launch boundaries between them; nothing is throughput-bound
        floor_us = None
        if spec.get("chain_floor", True):
            agent.close()
            del agent
            torch.cuda.empty_cache()
            tiny = build_config_agent(dict(spec, num_env=64))
            for _ in range(3):
                tiny.step()
            ups = []
            for _ in range(5):
                r2 = tiny.step()
                ups.append(r2.get("update_time", 0.0))
            floor_us = sorted(ups)[len(ups) // 2] / E * 1e6
            agent = tiny
        # kernel time of ONE policy epoch's dependent chain (the longer of
        # the two chains that run side by side) from the COMMITTED kernel trace
        # of this config (scripts/profiles_r05.sh -> scripts/rocpd_chain.py):
        # what is left of an epoch pair is launch boundary / dependency gap
        chain = None
        try:
            with open(os.path.join(REPO, "profiles",
                                   "r05_C4_bbrl_shard_chain.json")) as f:
                chain = json.load(f)
        except (OSError, ValueError):
            pass
        pair_us = upd / E * 1e6 if upd else None
        kern_us = chain["kernel_us_per_chain"] if chain and rows == 4096 \
            else None
        out["roofline"] = {
            "kernel": "critic + policy epochs (%d -> %d -> %d nets, %d rows)"
                      % (din, hs[0], hs[1], rows),
            "bound": "latency",
            "us_per_epoch_pair": round(pair_us, 1) if pair_us else None,
            "kernel_us_per_epoch_pair": kern_us,
            "gap_frac": None if not (kern_us and pair_us)
            else round(1.0 - kern_us / pair_us, 3),
            "chain": None if chain is None else {
                k: chain[k] for k in ("links", "kernel_us", "gap_us",
                                      "gap_to_next_chain_us")},
            "chain_source": None if chain is None
            else "profiles/r05_C4_bbrl_shard_chain.json",
            "chain_floor_us_per_epoch_pair": None if floor_us is None
            else round(floor_us, 1),
            "algorithmic_flops_per_epoch": flops,
            "note": "2 x %d dependent epochs of a few small kernels each: "
                    "launch / dependency latency, neither HBM nor MFMA.  "
                    "kernel_us_per_epoch_pair = sum of the kernel durations of "
                    "one policy epoch's chain (row kernel -> slab reduction -> "
                    "finish) in the committed trace of this config; gap_frac = "
                    "the share of an epoch pair that is launch boundary.  "
                    "chain_floor = the same update measured at 64 envs (one "
                    "workgroup per kernel)" % E}
    # every net of a BASELINE config runs on the hand-written kernels: a call
    # that reached library GEMMs would be a regression of the product path
    out["library_gemm_calls"] = sum(mlp_ops.LIBRARY_CALLS.values())
    assert not mlp_ops.LIBRARY_CALLS, \
        "config %s reached library GEMMs: %r" % (name, mlp_ops.LIBRARY_CALLS)
    agent.close()
    del agent
    torch.cuda.empty_cache()
    return out


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(full_size=False):
    """The CPU oracle (torch-CPU restatement of the reference path, kind
    'port') on a bounded sample: 64 envs (the reference's own cluster runs use
    16 - 38), full 50 + 50 epochs; three untimed warm-up steps, then the median
    of ten timed steps.  ~20 s of CPU work.

    full_size (``--cpu-full``): additionally ONE oracle step at the headline's
    4096 envs on the host cores of this run (BASELINE.md section 3's "same
    inputs in the same run"; minutes of CPU work, hence a flag) -- reported as
    ``full_size`` beside the sample."""
    from tce_rl_amd.config import tce_config
    from oracle.agent_oracle import OracleTCE      # checker / baseline only
    n = 64
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
    for _ in range(3):
        steps = o.step()                            # warm-up
    ts = []
    for _ in range(10):
        t = time.perf_counter()
        steps = o.step()
        ts.append(time.perf_counter() - t)
    dt = sorted(ts)[len(ts) // 2]
    # the same oracle at the HEADLINE size on every core, measured offline
    # (scripts/cpu_full_size.py: one step takes minutes) and committed
    full = ""
    try:
        with open(os.path.join(REPO, "profiles", "r05_cpu_4096.json")) as f:
            d = json.load(f)
        full = ("; the same oracle at the headline's 4096 envs on %d threads "
                "(%s; profiles/r05_cpu_4096.json, one step, offline): %.0f "
                "env-steps/s" % (d["cores"], d["cpu"], d["value"]))
    except (OSError, KeyError, ValueError):
        pass
    in_run = None
    if full_size:
        print("[bench] cpu_baseline: one oracle step at %d envs (minutes)"
              % NUM_ENV, file=sys.stderr, flush=True)
        cfg_f = tce_config("metaworld", num_env=NUM_ENV, num_basis=NUM_BASIS,
                           epochs=EPOCHS, device="cpu")
        of = OracleTCE(cfg_f["params"], NUM_ENV)
        t = time.perf_counter()
        steps_f = of.step()                         # (no warm-up: minutes each)
        dt_f = time.perf_counter() - t
        in_run = {"value": round(steps_f / dt_f, 1), "unit": "env-steps/s",
                  "cores": threads, "seconds_per_step": round(dt_f, 1),
                  "sample": "ONE torch-CPU oracle agent.step() at the "
                            "headline's %d envs (T 500, 50 + 50 epochs), no "
                            "warm-up, in this run" % NUM_ENV}
        del of
    return {"value": round(steps / dt, 1), "unit": "env-steps/s",
            "cores": threads, "kind": "port", "cpu": cpu_model(),
            "visible_cores": avail, "full_size": in_run,
            "sample": "torch-CPU oracle agent.step() at %d envs (T 500, 50 "
                      "critic + 50 policy epochs): 3 warm-up steps, median of "
                      "10 timed steps = %.2f s (min %.2f, max %.2f)%s"
                      % (n, dt, min(ts), max(ts), full)}


def exchange_report_over_ranks(agent, is_dist):
    """DistContext.exchange_report of every rank folded into one dict per
    channel: kind, self-test (all ranks), wait per collective (max over ranks
    of the mean and of the maximum, microseconds)."""
    if not agent.dist.active:
        return None
    from tce_rl_amd import dist as tdist
    mine = agent.dist.exchange_report()
    every = [mine]
    if is_dist and dist.get_world_size() > 1:
        every = [None] * dist.get_world_size()
        dist.all_gather_object(every, mine, group=tdist._boot_group())
    out = {}
    for ch in mine:
        rows = [e.get(ch, {}) for e in every]
        rep = {"kind": mine[ch]["kind"]}
        if mine[ch]["kind"] == "xgmi-oneshot":
            means = [r["wait_us_mean"] for r in rows
                     if r.get("wait_us_mean") is not None]
            rep.update({
                "self_test": all(r.get("self_test") in (True, None)
                                 for r in rows),
                "collectives": mine[ch]["collectives"],
                "wait_us_mean_max_over_ranks": max(means) if means else None,
                "wait_us_max_over_ranks": max(
                    r.get("wait_us_max", 0.0) for r in rows)})
        out[ch] = rep
    return out


def _kill_tree(proc):
    """End the launcher child and everything it started (it runs in its own
    session, so its process group holds exactly its descendants)."""
    import signal
    for sig in (signal.SIGTERM, signal.SIGKILL):
        try:
            os.killpg(proc.pid, sig)
        except (ProcessLookupError, PermissionError):
            return
        try:
            proc.wait(timeout=10)
            return
        except Exception:
            continue


def _launch_attempts(base_env):
    """The environments self_launch tries, in order: the
    job as configured; the other HIP IPC mode (the in-library exchange and
    RCCL both map peer memory through it, and which one a host driver supports
    is the one assumption no one-GPU box can test); the configured IPC mode
    with the gradients on torch.distributed all-reduces instead of the
    in-library exchange."""
    ipc = base_env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    other = "1" if ipc == "0" else "0"
    return [("as configured", {}),
            ("HSA_ENABLE_IPC_MODE_LEGACY=%s" % other,
             {"HSA_ENABLE_IPC_MODE_LEGACY": other}),
            ("TCE_EXCHANGE=rccl", {"TCE_EXCHANGE": "rccl"})]


def self_launch(args):
    """``python bench.py --gpus N`` from a plain shell (WORLD_SIZE unset): this
    process has not touched the GPU and never will -- it starts the N ranks as
    fresh children through torch.distributed.run (no exec), relays rank 0's
    JSON line and returns the children's exit code.

    A child set that fails (non-zero exit, or no "warmup done" within
    ``--startup-timeout`` seconds) BEFORE rank 0 has finished its warm-up steps
    is replaced ONCE per entry of ``_launch_attempts`` by a fresh child set
    with the next environment -- new processes, never a re-exec; the line
    records which attempt produced it (``launch_attempt``).  A failure after
    the warm-up is a real failure and is returned.  All attempts together get
    ``--launch-timeout`` seconds (default 900: a hung collective must not sit
    until the driver's limit); past it the child tree is killed and the exit
    code is 124."""
    import socket
    import subprocess
    import tempfile
    base_cmd = ["--gpus", str(args.gpus), "--steps", str(args.steps),
                "--warmup", str(args.warmup)]
    if args.no_cpu_baseline:
        base_cmd.append("--no-cpu-baseline")
    if args.with_split_f16:
        base_cmd.append("--with-split-f16")
    if getattr(args, "critic_arith", "f32") != "f32":
        base_cmd += ["--critic-arith", args.critic_arith]
    if getattr(args, "no_configs", False):
        base_cmd.append("--no-configs")
    if getattr(args, "scaling", "weak") != "weak":
        base_cmd += ["--scaling", args.scaling]
    base_env = dict(os.environ)
    base_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base_env.setdefault("OMP_NUM_THREADS", "4")
    limit = float(getattr(args, "launch_timeout", 900.0))
    startup = float(getattr(args, "startup_timeout", 300.0))
    deadline = time.monotonic() + limit
    attempts = _launch_attempts(base_env)
    if os.environ.get("TCE_BENCH_NO_RETRY") == "1":
        attempts = attempts[:1]
    rc, line = 1, None
    scratch = tempfile.mkdtemp(prefix="tce_bench_")
    for no, (label, extra) in enumerate(attempts, 1):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               "--nproc-per-node", str(args.gpus), "--master-addr",
               "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + base_cmd
        sentinel = os.path.join(scratch, "warm_%d" % no)
        env = dict(base_env)
        env.update(extra)
        env["TCE_BENCH_SENTINEL"] = sentinel
        env["TCE_BENCH_LAUNCH_ATTEMPT"] = "%d: %s" % (no, label)
        out_path = os.path.join(scratch, "stdout_%d" % no)
        timed_out = False
        with open(out_path, "wb") as out_f:
            proc = subprocess.Popen(cmd, env=env, stdout=out_f,
                                    start_new_session=True)
            t_start = time.monotonic()
            try:
                while proc.poll() is None:
                    now = time.monotonic()
                    warm = os.path.exists(sentinel)
                    if now > deadline or (not warm and
                                          now - t_start > startup):
                        timed_out = True
                        print("[bench] attempt %d (%s): the %d ranks %s: "
                              "killing the child tree" % (
                                  no, label, args.gpus,
                                  "did not finish within %.0f s" % limit
                                  if now > deadline else
                                  "had not finished their warm-up after "
                                  "%.0f s" % startup),
                              file=sys.stderr, flush=True)
                        _kill_tree(proc)
                        break
                    time.sleep(0.2)
            except BaseException:
                _kill_tree(proc)
                raise
            rc = 124 if timed_out else proc.returncode
        line = None
        with open(out_path, "rb") as f:
            for ln in f.read().decode(errors="replace").splitlines():
                if ln.startswith("{") and '"metric"' in ln:
                    line = ln
                else:
                    print(ln, file=sys.stderr)
        warm = os.path.exists(sentinel)
        if rc == 0 or warm or time.monotonic() > deadline:
            break
        if no < len(attempts):
            print("[bench] attempt %d (%s) ended with exit code %d before the "
                  "warm-up was done: starting a fresh child set (%s)"
                  % (no, label, rc, attempts[no][0]), file=sys.stderr,
                  flush=True)
    import shutil
    shutil.rmtree(scratch, ignore_errors=True)
    if line is not None and rc == 0:
        print(line, flush=True)
    elif rc == 0:
        print("[bench] no JSON line from rank 0", file=sys.stderr)
        return 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-full", action="store_true",
                    help="cpu_baseline additionally times ONE oracle step at "
                         "the headline's 4096 envs on this run's host cores "
                         "(~3.5 min; BASELINE.md section 3's definition)")
    ap.add_argument("--with-split-f16", action="store_true",
                    help="add a second timed region with the agent option "
                         "critic_arith=f16x2 (split-f16 critic kernel: an "
                         "option, not the reported value)")
    ap.add_argument("--critic-arith", choices=["f32", "bf16x3", "f16x2"],
                    default="f32",
                    help="arithmetic of the critic epochs of the TIMED steps "
                         "(default f32 = the exact-fp32 matrix instructions, "
                         "what `value` is quoted on; bf16x3 = three-part bf16 "
                         "operands, 24 bits, csrc/mlpb.hip: the line then says "
                         "so in `dtype` and `critic_arith`)")
    ap.add_argument("--no-split-f16", action="store_true",
                    help="accepted for older command lines (the default now)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` block (the other BASELINE.json "
                         "configs, a few timed steps each)")
    ap.add_argument("--config-steps", type=int, default=3)
    ap.add_argument("--config-warmup", type=int, default=4)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default, what the driver's curve uses): 4096 "
                         "envs per GPU; strong: 4096 envs in total, "
                         "4096 / N per GPU (SURVEY 8d)")
    ap.add_argument("--launch-timeout", type=float, default=900.0,
                    help="seconds the self-launched ranks may take")
    ap.add_argument("--startup-timeout", type=float, default=300.0,
                    help="seconds a self-launched child set may take until "
                         "rank 0 has finished its warm-up steps; past it the "
                         "set is killed and the next launch attempt starts")
    ap.add_argument("--collective-timeout", type=float, default=120.0,
                    help="seconds a collective may take before the process "
                         "group aborts (a hang must not sit until the "
                         "driver's limit)")
    args = ap.parse_args()
    # before anything initialises the HIP / HSA runtime (dmabuf IPC for RCCL)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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
        import datetime
        tmo = datetime.timedelta(seconds=args.collective_timeout)
        if backend == "nccl":
            dist.init_process_group("nccl", timeout=tmo,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, timeout=tmo)

    # the GLOBAL env count: MPExperiment gives every rank its share of them
    # and the env / noise seed `seed + rank`.  weak: 4096 per GPU; strong:
    # 4096 in total
    strong = args.scaling == "strong"
    if strong and NUM_ENV % world:
        raise SystemExit("bench.py --scaling strong: %d envs do not divide "
                         "over %d ranks" % (NUM_ENV, world))
    envs_per_rank = NUM_ENV // world if strong else NUM_ENV
    agent, cfg = build_agent(envs_per_rank * world, seed=0)
    agent.critic_arith = args.critic_arith
    assert agent.sampler.num_env_train == envs_per_rank
    T = agent.sampler.num_times

    is_dist = dist.is_initialized()

    from tce_rl_amd import dist as tdist

    def barrier():
        """Every rank's device work done, then a barrier over the ranks (on the
        host -- a gloo twin of the RCCL group -- so that the barrier itself
        puts nothing on the GPU), then the device once more."""
        torch.cuda.synchronize()
        if is_dist:
            tdist.host_barrier()
        torch.cuda.synchronize()

    def over_ranks(vals):
        """max over ranks of each value, plus every rank's first value."""
        if not is_dist:
            return vals, [vals[0]]
        every = [None] * dist.get_world_size()
        dist.all_gather_object(every, [float(v) for v in vals],
                               group=tdist._boot_group())
        return [max(e[i] for e in every) for i in range(len(vals))], \
            [e[0] for e in every]

    # (rehearsal hook of the launcher's retry path, tests/test_bench_dist_gpu.py:
    # TCE_BENCH_DIE_BEFORE_WARMUP="1,2" makes the ranks of launch attempts 1 and
    # 2 exit with code 3 here -- after the process group and the exchanges are
    # up, before the warm-up -- as a child set would that cannot map its peers)
    die = os.environ.get("TCE_BENCH_DIE_BEFORE_WARMUP", "")
    if die and os.environ.get("TCE_BENCH_LAUNCH_ATTEMPT", "0")[0] in \
            die.split(","):
        print("[bench] rank %d: dying before the warm-up (rehearsal)" % rank,
              file=sys.stderr, flush=True)
        os._exit(3)
    for _ in range(args.warmup):
        agent.step()
    barrier()
    if rank == 0:
        print("[bench] warmup done", file=sys.stderr, flush=True)
        if os.environ.get("TCE_BENCH_SENTINEL"):
            # (self_launch: a failure from here on is not a launch problem)
            open(os.environ["TCE_BENCH_SENTINEL"], "w").close()
    agent.dist.check_exchanges()
    tdist.reset_stats()
    agent.dist.reset_wait_stats()
    # marks in the kernel trace around the timed steps (scripts/rocpd_stats.py
    # --between-markers 1 2: the committed profile covers exactly this window)
    from tce_rl_amd import _lib as tlib
    tlib.call("tce_marker", 1, tlib.stream())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pol_time = crit_time = 0.0
    results = []
    for _ in range(args.steps):
        # (the agent returns its metrics as a mapping that is filled on first
        # access -- agent.lazy_metrics -- and they are read after the barrier:
        # the host enqueues the next rollout behind the critic epochs)
        results.append(agent.step())
    barrier()
    elapsed = time.perf_counter() - t0
    tlib.call("tce_marker", 2, tlib.stream())
    for res in results:
        pol_time += res["update_policy_time"]
        crit_time += res["update_critic_time"]      # device time (HIP events)
    agent.dist.check_exchanges()        # (a wait that ran into its limit is fatal)
    coll = tdist.stats()                # torch.distributed + in-library collectives
    (elapsed, pol_time), per_rank = over_ranks([elapsed, pol_time])
    xreport = exchange_report_over_ranks(agent, is_dist)
    # one iteration in `balance_check` (25) carries the policy balance check:
    # timed by itself, outside the K steps
    bal_ms = None
    if isinstance(agent.balance_check, int):
        bal_ms = over_ranks([time_balance_iteration(agent, barrier)])[0][0]

    # the same K steps (after W warm-up steps) with the critic epochs on the
    # split-f16 kernel (agent option critic_arith="f16x2"), reported beside
    # the fp32 figure as "split_f16_critic"
    fast = None
    if args.with_split_f16:
        agent.critic_arith = "f16x2"
        agent._critic_split = 0
        for _ in range(args.warmup):
            agent.step()
        barrier()
        t1 = time.perf_counter()
        pol16 = 0.0
        res16 = []
        for _ in range(args.steps):
            res16.append(agent.step())
        barrier()
        el16 = time.perf_counter() - t1
        for res in res16:
            pol16 += res["update_policy_time"]
        fast, _ = over_ranks([el16, pol16])
        agent.critic_arith = args.critic_arith

    if rank == 0:
        env_steps = world * envs_per_rank * T * args.steps
        print("[bench] timed region: %.3f s" % elapsed, file=sys.stderr,
              flush=True)
        roof, extra = roofline(agent, crit_time / args.steps * 1e3,
                               args.with_split_f16, envs_per_rank)
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
            # the K timed steps are ordinary iterations; the reference runs its
            # balance check in 1 of `balance_check` iterations (25 in every
            # shipped YAML): that iteration alone, and the average over a cycle
            "balance_check_iteration_ms": None if bal_ms is None
            else round(bal_ms, 2),
            "ms_per_step_amortised": None if bal_ms is None else round(
                ((agent.balance_check - 1) * elapsed / args.steps * 1e3 +
                 bal_ms) / agent.balance_check, 2),
            "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32" if args.critic_arith == "f32" else
            "f32 (critic epochs: %s operands, fp32 accumulate)" % args.critic_arith,
            "critic_arith": args.critic_arith, "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: TCE, Metaworld-"
                       "reach-like synthetic env, 4096 envs per GPU, T 500, "
                       "P 24, dof 4, ProDMP 5 basis (K 24), 50 critic + 50 "
                       "policy epochs, KL projection, fp32",
                       "num_env_per_gpu": envs_per_rank, "num_times": T,
                       "num_basis": NUM_BASIS, "epochs": EPOCHS,
                       "parallelism": "env-shard x%d" % world},
            "backend": (dist.get_backend() if is_dist else None),
            "rccl_ranks": (dist.get_world_size() if is_dist and
                           dist.get_backend() == "nccl" else 0),
            "ms_per_step_per_rank": [round(t / args.steps * 1e3, 2)
                                     for t in per_rank],
            # what this rank put on the wire per step (tce_rl_amd/dist.py
            # counters over the timed region; all ranks issue the same)
            # what carries the gradients: "xgmi-oneshot" = the in-library
            # exchange inside the finish kernels (csrc/xchg.h), "rccl" =
            # torch.distributed all-reduces between the C calls
            "gradient_exchange": agent.dist.exchange_kind(),
            # per channel (critic / policy gradients, the small per-step
            # collectives): the start-up self-test's result and how long
            # workgroup 0 of a collective waited for its slowest peer
            # (device clock inside xchg_sync, csrc/xchg.h) -- mean per
            # collective and maximum, each the MAX over ranks; what a
            # straggler costs is readable from this line alone
            "exchange": xreport,
            "launch_attempt": os.environ.get("TCE_BENCH_LAUNCH_ATTEMPT",
                                             "driver-launched"),
            "hsa_ipc_mode_legacy": os.environ.get(
                "HSA_ENABLE_IPC_MODE_LEGACY"),
            "collectives_per_step": round(coll["collectives"] / args.steps, 1),
            "collective_bytes_per_step": round(coll["bytes"] / args.steps),
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
        if world == 1 and not args.no_configs:
            agent.close()
            del agent
            torch.cuda.empty_cache()
            out["configs"] = {}
            for name, spec in OTHER_CONFIGS:
                out["configs"][name] = run_config(
                    name, spec, args.config_steps, args.config_warmup)
                print("[bench] config %s: %.1f ms/step" % (
                    name, out["configs"][name]["ms_per_step"]),
                    file=sys.stderr, flush=True)
            b3 = out["configs"].get("C2_bf16x3_critic")
            if b3 and args.critic_arith == "f32":
                # the same workload with the fp32-grade critic epochs on the bf16 matrix
                # cores: beside `value` (exact-fp32 instructions), not instead of it
                out["with_critic_arith_bf16x3"] = {
                    "value": b3["env_steps_per_sec"], "unit": "env-steps/s",
                    "ms_per_step": b3["ms_per_step"],
                    "policy_updates_per_sec": b3.get("policy_updates_per_sec"),
                    "see": "configs.C2_bf16x3_critic, roofline_extra.critic_bf16x3; "
                           "`bench.py --critic-arith bf16x3` times the K steps of the "
                           "headline with it (profiles/r05_bench_line_bf16x3.json)",
                    "operands": "x = b0 + b1 + b2 exactly (3 bf16 parts: 24 bits, fp32 "
                                "range), six partial products, fp32 accumulate; gradient "
                                "error vs fp64 at C2: 1.65e-6 (exact-fp32 kernel: 2.02e-6; "
                                "tests/test_mlpb_gpu.py)"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_full)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if is_dist:
        if "agent" in locals():
            agent.close()               # (collective: buffers, IPC mappings)
        tdist.host_barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
