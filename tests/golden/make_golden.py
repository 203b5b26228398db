"""Generate the golden vectors in this directory from the reference itself.

Runs ONLY in the build container (needs ``/root/reference``); the ``.npz``
fixtures it writes are committed, the reference never travels.  Third-party
modules the reference imports but that are not installed (wandb, cw2,
mp_pytorch, trust_region_projections, ...) are replaced by inert
``MagicMock`` modules before import; only mprl-owned
plain-torch functions are executed.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True

_STUBS = [
    "wandb", "cw2", "cw2.cluster_work", "cw2.cw_data",
    "cw2.cw_data.cw_wandb_logger", "cw2.cw_data.cw_logging", "cw2.experiment",
    "cw2.cw_error", "natsort", "git_repos_tracker",
    "git_repos_tracker.tracker", "mp_pytorch", "mp_pytorch.basis_gn",
    "mp_pytorch.mp", "mp_pytorch.phase_gn", "stable_baselines3",
    "stable_baselines3.common", "stable_baselines3.common.vec_env",
    "fancy_gym", "gymnasium", "trust_region_projections",
    "trust_region_projections.utils",
    "trust_region_projections.utils.projection_utils",
    "trust_region_projections.projections",
    "trust_region_projections.projections.base_projection_layer",
    "trust_region_projections.projections.frob_projection_layer",
    "trust_region_projections.projections.kl_projection_layer",
    "trust_region_projections.projections.papi_projection",
    "trust_region_projections.projections.w2_projection_layer",
    "trust_region_projections.projections.w2_projection_layer_non_com",
]


def import_reference():
    for n in _STUBS:
        sys.modules[n] = MagicMock()

    class _Empty:
        pass

    sys.modules["cw2.experiment"].AbstractExperiment = _Empty
    sys.modules["cw2.experiment"].AbstractIterativeExperiment = _Empty
    sys.path.insert(0, "/root/reference")
    import mprl.util as util
    from mprl.rl.agent import TemporalCorrelatedAgent, BlackBoxAgent
    from mprl.rl.policy import BlackBoxPolicy, TemporalCorrelatedPolicy
    from mprl.rl.critic import ValueFunction
    return util, TemporalCorrelatedAgent, BlackBoxAgent, BlackBoxPolicy, \
        TemporalCorrelatedPolicy, ValueFunction


def npy(d):
    out = {}
    for k, v in d.items():
        out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    return out


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **npy(arrays))
    print("wrote", path, len(arrays), "arrays")


def main():
    util, TCAgent, BBAgent, BBPolicy, TCPolicy, ValueFunction = \
        import_reference()
    from mprl.rl.sampler.temporal_correlated_sampler import \
        TemporalCorrelatedSampler as TCS
    g = torch.Generator().manual_seed(1234)
    rn = lambda *s, dtype=torch.float32: torch.randn(*s, generator=g, dtype=dtype)
    ru = lambda *s, dtype=torch.float32: torch.rand(*s, generator=g, dtype=dtype)

    # (1) pair selection: bit-exact incl. the RNG call ------------------
    d = {}
    for T in (100, 250, 350, 500):
        for s in range(10):
            torch.manual_seed(s)
            d[f"fixed_T{T}_s{s}"] = util.select_pred_pairs(
                num_all=T, num_select=25, fixed_interval=True).to(torch.long)
            # the value drawn right after: pins the generator position
            d[f"fixed_T{T}_s{s}_next"] = torch.randint(0, 1 << 30, size=[])
    for s in range(3):
        torch.manual_seed(s)
        d[f"random_T100_s{s}"] = util.select_pred_pairs(
            num_all=100, num_select=25, fixed_interval=False).to(torch.long)
    save("pred_pairs", **d)

    # (2) time grid --------------------------------------------------------
    d = {}
    for i, (dt, T) in enumerate(((0.0125, 500), (0.02, 100), (0.008, 350))):
        sampler = types.SimpleNamespace(dt=dt)
        t0 = torch.cat([torch.zeros(3), ru(5)])
        d[f"t0_{i}"], d[f"dt_{i}"], d[f"T_{i}"] = t0, dt, T
        d[f"times_{i}"] = TCS.get_times(sampler, t0, T)
    save("times", **d)

    # (3) GAE ------------------------------------------------------------------
    d = {}
    case = 0
    for dtype in (torch.float32, torch.float64):
        for gamma in (1.0, 0.99):
            for use_gae in (True, False):
                N, T = 12, 100
                r, v = rn(N, T, dtype=dtype), rn(N, T + 1, dtype=dtype)
                dones = ru(N, T) < 0.03
                dones[:, -1] = True
                tl = ru(N, T) < 0.02 if case % 3 == 2 else \
                    torch.zeros(N, T, dtype=torch.bool)
                ag = types.SimpleNamespace(
                    discount_factor=torch.tensor(gamma, dtype=dtype),
                    use_gae=use_gae, gae_scaling=0.95)
                adv, ret = TCAgent.get_advantage_return(ag, r, v, dones, tl)
                d.update({f"r_{case}": r, f"v_{case}": v, f"dones_{case}": dones,
                          f"tl_{case}": tl, f"gamma_{case}": gamma,
                          f"use_gae_{case}": use_gae, f"adv_{case}": adv,
                          f"ret_{case}": ret})
                case += 1
    d["num_cases"] = case
    save("gae", **d)

    # (4) segment advantage ----------------------------------------------------
    d = {}
    case = 0
    for dtype in (torch.float32, torch.float64):
        for mode in ("value_subtraction", "accumulate", "accumulated_rewards"):
            for norm, clip, gamma in ((True, 0.0, 1.0), (False, 0.0, 0.99),
                                      (True, 1.5, 0.99)):
                N, T = 16, 100
                r, v, a = rn(N, T, dtype=dtype), rn(N, T + 1, dtype=dtype), \
                    rn(N, T, dtype=dtype)
                torch.manual_seed(case)
                pairs = util.select_pred_pairs(
                    num_all=T, num_select=25, fixed_interval=True).to(torch.long)
                ag = types.SimpleNamespace(
                    discount_factor=torch.tensor(gamma, dtype=dtype),
                    segment_advantage=mode, norm_advantages=norm,
                    clip_advantages=clip, dtype=dtype,
                    device=torch.device("cpu"))
                out = TCAgent.get_segment_advantage(ag, r, v, a, pairs)
                d.update({f"r_{case}": r, f"v_{case}": v, f"a_{case}": a,
                          f"pairs_{case}": pairs, f"mode_{case}": mode,
                          f"norm_{case}": norm, f"clip_{case}": clip,
                          f"gamma_{case}": gamma, f"out_{case}": out})
                case += 1
    d["num_cases"] = case
    save("segment_advantage", **d)

    # (5) Cholesky head round trip ----------------------------------------------
    d = {}
    for K in (20, 24, 28, 36, 63):
        for std_only in (False, True):
            pol = types.SimpleNamespace(dim_out=K, std_only=std_only,
                                        min_std=1e-5)
            n = K if std_only else K + K * (K - 1) // 2
            vec = rn(3, n)
            L = BBPolicy._vector_to_cholesky(pol, vec)
            back = BBPolicy._cholesky_to_vector(pol, L)
            tag = f"K{K}_{'diag' if std_only else 'full'}"
            d[f"vec_{tag}"], d[f"L_{tag}"], d[f"back_{tag}"] = vec, L, back
    d["init_var_full_K24"] = util.reverse_from_softplus_space(
        torch.ones(24), lower_bound=None)
    d["softplus_known"] = torch.stack([
        util.to_softplus_space(torch.tensor(0.0), None),
        util.to_softplus_space(torch.tensor(0.0), 2)])
    save("cholesky_head", **d)

    # (6) param-space Gaussian + autograd grads ------------------------------------
    d = {}
    for K in (20, 36):
        N = 6
        mean = rn(N, K).requires_grad_(True)
        vec = rn(N, K + K * (K - 1) // 2) * 0.3
        pol = types.SimpleNamespace(dim_out=K, std_only=False, min_std=1e-5)
        L = BBPolicy._vector_to_cholesky(pol, vec).detach().requires_grad_(True)
        eps = rn(N, K)
        x = (mean + torch.einsum('nij,nj->ni', L, eps)).detach()
        other = rn(N, K)
        lp = BBPolicy.log_prob(None, x, mean, L)
        ent = BBPolicy.entropy(None, [mean, L])
        w = rn(N)
        (lp * w).sum().backward()
        d.update({f"mean_K{K}": mean, f"L_K{K}": L, f"eps_K{K}": eps,
                  f"x_K{K}": x, f"other_K{K}": other, f"w_K{K}": w,
                  f"logp_K{K}": lp, f"ent_K{K}": ent,
                  f"dmean_K{K}": mean.grad.clone(), f"dL_K{K}": L.grad.clone(),
                  f"cov_K{K}": BBPolicy.covariance(None, L),
                  f"logdet_K{K}": BBPolicy.log_determinant(None, L),
                  f"prec_K{K}": BBPolicy.precision(None, L.detach()),
                  f"maha_K{K}": BBPolicy.maha(None, mean, other, L)})
    save("mvn", **d)

    # (7) MLP / losses / grad-norm ------------------------------------------------------
    d = {}
    for i, act in enumerate(("tanh", "relu", "leaky_relu", "softplus")):
        torch.manual_seed(10 + i)
        crit = ValueFunction(dim_in=11, dim_out=1,
                             hidden={"avg_neuron": 16, "num_hidden": 2,
                                     "shape": 0.0},
                             init_method="orthogonal", out_layer_gain=1.0,
                             act_func_hidden=act, act_func_last=None,
                             dtype="float32", device="cpu")
        x = rn(7, 5, 11)
        sd = crit.net.state_dict()
        for j, (k, v) in enumerate(sd.items()):
            d[f"{act}_p{j}"] = v
        d[f"{act}_x"], d[f"{act}_y"] = x, crit.critic(x)
        d[f"{act}_seed"] = 10 + i
    vals, rets, old = rn(50), rn(50), rn(50)
    ag = types.SimpleNamespace(clip_critic=0.0)
    d["vl_values"], d["vl_returns"], d["vl_old"] = vals, rets, old
    d["vl_unclipped"] = TCAgent.value_loss(ag, vals, rets, old)
    ag.clip_critic = 0.2
    d["vl_clipped"] = TCAgent.value_loss(ag, vals, rets, old)
    adv, lpn, lpo = rn(9, 24), rn(9, 24) * 0.1, rn(9, 24) * 0.1
    sl, st = TCAgent.surrogate_loss(adv, lpn, lpo)
    d.update(sl_adv=adv, sl_new=lpn, sl_old=lpo, sl_loss=sl,
             sl_ratio=st["imp_smp_ratio"])
    ps = [torch.nn.Parameter(rn(4, 3)), torch.nn.Parameter(rn(5))]
    gs = [rn(4, 3), rn(5)]
    for p, gg in zip(ps, gs):
        p.grad = gg.clone()
    before, after = util.grad_norm_clip(0.5, ps)
    d.update(gn_g0=gs[0], gn_g1=gs[1], gn_before=before, gn_after=after,
             gn_c0=ps[0].grad, gn_c1=ps[1].grad)
    d["arch_128_2_0"] = util.mlp_arch_3_params(128, 2, 0.0)
    d["arch_64_3_m05"] = util.mlp_arch_3_params(64, 3, -0.5)
    d["arch_256_1_0"] = util.mlp_arch_3_params(256, 1, 0.0)
    save("mlp_losses", **d)

    # (8) running mean / std after three updates ---------------------------------------------
    rms = util.RunningMeanStd(name="obs", shape=(6,), dtype="float32",
                              device="cpu")
    d = {}
    for i in range(3):
        arr = rn(40 + 10 * i, 6) * (1 + i) + i
        rms.update(arr)
        d[f"arr_{i}"] = arr
    d.update(mean=rms.mean, var=rms.var, count=rms.count)
    save("rms", **d)

    # (9) mdp reward ---------------------------------------------------------------
    N, T = 8, 30
    r = rn(N, T)
    first = torch.tensor([5, 0, -1, 29, 12, 1, -1, 20])   # -1: never
    flags = torch.zeros(N, T, dtype=torch.bool)
    for n in range(N):
        if first[n] >= 0:
            flags[n, first[n]:] = True
    infos = [{"hit_ball": flags[n].numpy()} for n in range(N)]
    out = util.make_mdp_reward("fancy_ProDMP_TCE/TableTennisRndInit-v0",
                               r.clone(), infos, torch.float32,
                               torch.device("cpu"))
    same = util.make_mdp_reward("metaworld_ProDMP_TCE/reach-v2", r.clone(),
                                infos, torch.float32, torch.device("cpu"))
    save("mdp_reward", r=r, flags=flags, out=out, noop=same)

    # (10) pair-wise log-prob index plumbing with the build's ProDMP injected ------------------------
    sys.path.insert(0, REPO)
    from oracle.prodmp_oracle import ProDMPOracle

    class MPAdapter:
        """Speaks the mp_pytorch surface the reference calls, on the oracle."""

        def __init__(self, mp):
            self.o, self.num_dof = mp, mp.num_dof

        def update_inputs(self, times=None, params=None, params_L=None,
                          init_time=None, init_pos=None, init_vel=None):
            self.a = (times, params, params_L, init_time, init_pos, init_vel)

        def get_traj_pos(self, flat_shape=False, **kw):
            t, p, _, t0, p0, v0 = self.a
            if flat_shape:
                return self.o.traj_pos_flat(t, p, t0, p0, v0)
            return self.o.traj(t, p, t0, p0, v0)[0]

        def get_traj_pos_cov(self):
            t, _, L, t0, _, _ = self.a
            return self.o.traj_pos_cov(t, L, t0)

    d = {}
    for tag, dof, nb, T, dt, tau in (("mw", 4, 8, 500, 0.0125, 5.0),
                                     ("bp", 7, 8, 100, 0.02, 2.0)):
        cfg = dict(num_dof=dof, num_basis=nb, tau=tau, alpha_phase=3, alpha=10,
                   dt=dt, basis_bandwidth_factor=5 if tag == "mw" else 3,
                   weights_scale=0.1 if tag == "mw" else 0.3,
                   goal_scale=0.1 if tag == "mw" else 0.3, relative_goal=True)
        mp = ProDMPOracle(dtype=torch.float32, **cfg)
        K = dof * (nb + 1)
        N = 5
        pol = types.SimpleNamespace(mp=MPAdapter(mp), num_dof=dof)
        t0 = torch.zeros(N)
        sampler = types.SimpleNamespace(dt=dt)
        times = TCS.get_times(sampler, t0, T)
        mean = rn(N, K) * 0.5
        hp = types.SimpleNamespace(dim_out=K, std_only=False, min_std=1e-5)
        L = BBPolicy._vector_to_cholesky(
            hp, torch.cat([rn(N, K), 0.05 * rn(N, K * (K - 1) // 2)], -1))
        eps = rn(N, K)
        y0, v0 = ru(N, dof) * 2 - 1, 0.1 * rn(N, dof)
        pos, vel = mp.sample_trajectories(times, mean, L, t0, y0, v0, eps)
        traj = torch.cat([pos, vel], -1)
        torch.manual_seed(3)
        pairs = util.select_pred_pairs(num_all=T, num_select=25,
                                       fixed_interval=True).to(torch.long)
        lp = TCPolicy.log_prob(pol, traj, mean, L, times, t0, y0, v0,
                               pred_pairs=pairs)
        d.update({f"{tag}_traj": traj, f"{tag}_mean": mean, f"{tag}_L": L,
                  f"{tag}_eps": eps, f"{tag}_times": times, f"{tag}_t0": t0,
                  f"{tag}_y0": y0, f"{tag}_v0": v0, f"{tag}_pairs": pairs,
                  f"{tag}_logp": lp})
        for k, v in cfg.items():
            d[f"{tag}_cfg_{k}"] = v
    save("pair_logprob_plumbing", **d)


if __name__ == "__main__":
    main()
