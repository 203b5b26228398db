"""The CPU oracle against the golden vectors generated from the reference
(tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import torch

from oracle import tce_oracle as O
from oracle.prodmp_oracle import ProDMPOracle, pair_log_prob

T = torch.as_tensor


def test_pred_pairs_bit_exact(golden):
    g = golden("pred_pairs")
    for num_all in (100, 250, 350, 500):
        for s in range(10):
            torch.manual_seed(s)
            pairs = O.get_time_pairs(num_all, dict(num_select=25,
                                                   fixed_interval=True))
            assert pairs.dtype == torch.long
            assert np.array_equal(pairs.numpy(), g[f"fixed_T{num_all}_s{s}"])
            # generator position after the call
            nxt = torch.randint(0, 1 << 30, size=[]).item()
            assert nxt == int(g[f"fixed_T{num_all}_s{s}_next"])
    for s in range(3):
        torch.manual_seed(s)
        pairs = O.get_time_pairs(100, dict(num_select=25, fixed_interval=False))
        assert np.array_equal(pairs.numpy(), g[f"random_T100_s{s}"])


def test_times_bit_exact(golden):
    g = golden("times")
    for i in range(3):
        out = O.get_times(T(g[f"t0_{i}"]), float(g[f"dt_{i}"]),
                          int(g[f"T_{i}"]))
        assert np.array_equal(out.numpy(), g[f"times_{i}"])


def test_gae_bit_exact(golden):
    g = golden("gae")
    for c in range(int(g["num_cases"])):
        adv, ret = O.gae(T(g[f"r_{c}"]), T(g[f"v_{c}"]), T(g[f"dones_{c}"]),
                         T(g[f"tl_{c}"]), float(g[f"gamma_{c}"]), 0.95,
                         bool(g[f"use_gae_{c}"]))
        assert np.array_equal(adv.numpy(), g[f"adv_{c}"])
        assert np.array_equal(ret.numpy(), g[f"ret_{c}"])


def test_segment_advantage(golden):
    g = golden("segment_advantage")
    for c in range(int(g["num_cases"])):
        out = O.segment_advantage(str(g[f"mode_{c}"]), T(g[f"r_{c}"]),
                                  T(g[f"v_{c}"]), T(g[f"a_{c}"]),
                                  T(g[f"pairs_{c}"]), float(g[f"gamma_{c}"]),
                                  bool(g[f"norm_{c}"]), float(g[f"clip_{c}"]))
        np.testing.assert_allclose(out.numpy(), g[f"out_{c}"], rtol=1e-6,
                                   atol=1e-6)


def test_cholesky_head(golden):
    g = golden("cholesky_head")
    for K in (20, 24, 28, 36, 63):
        for std_only in (False, True):
            tag = f"K{K}_{'diag' if std_only else 'full'}"
            L = O.vector_to_cholesky(T(g[f"vec_{tag}"]), K, 1e-5, std_only)
            assert np.array_equal(L.numpy(), g[f"L_{tag}"])
            back = O.cholesky_to_vector(L, 1e-5, std_only)
            assert np.array_equal(back.numpy(), g[f"back_{tag}"])
    v = O.initial_variance_vector(24, False)
    assert np.array_equal(v[:24].numpy(), g["init_var_full_K24"])
    assert float(v[24:].abs().max()) == 0
    # known answers printed by the reference's own test file
    # (mprl/test/util_test/util_numerical_test.py:21-36)
    np.testing.assert_allclose(g["softplus_known"], [0.7031, 2.6931],
                               atol=1e-4)
    assert abs(float(O.reverse_from_softplus_space(
        torch.tensor(0.70314717), None))) < 1e-6


def test_build_lower_matrix_known_answer():
    # mprl/test/util_test/util_matrix_test.py:9-35: row-major tril fill
    diag = torch.full((6,), 0.5)
    off = torch.arange(1.0, 16.0)
    L = O.build_lower_matrix(diag, off)
    assert L[1, 0] == 1 and L[2, 0] == 2 and L[2, 1] == 3 and L[5, 4] == 15
    d2, o2 = O.reverse_build_matrix(L, True)
    assert torch.equal(d2, diag) and torch.equal(o2, off)


def test_mvn(golden):
    g = golden("mvn")
    for K in (20, 36):
        mean = T(g[f"mean_K{K}"]).requires_grad_(True)
        L = T(g[f"L_K{K}"]).requires_grad_(True)
        x = O.mvn_rsample(mean, L, T(g[f"eps_K{K}"])).detach()
        np.testing.assert_allclose(x.numpy(), g[f"x_K{K}"], rtol=1e-6,
                                   atol=1e-6)
        lp = O.mvn_log_prob(T(g[f"x_K{K}"]), mean, L)
        np.testing.assert_allclose(lp.detach().numpy(), g[f"logp_K{K}"],
                                   rtol=1e-6)
        (lp * T(g[f"w_K{K}"])).sum().backward()
        np.testing.assert_allclose(mean.grad.numpy(), g[f"dmean_K{K}"],
                                   rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(L.grad.numpy(), g[f"dL_K{K}"], rtol=1e-5,
                                   atol=1e-6)
        Ld = L.detach()
        np.testing.assert_allclose(O.mvn_entropy(mean.detach(), Ld).numpy(),
                                   g[f"ent_K{K}"], rtol=1e-6)
        np.testing.assert_allclose(O.covariance(Ld).numpy(), g[f"cov_K{K}"],
                                   rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(O.log_determinant(Ld).numpy(),
                                   g[f"logdet_K{K}"], rtol=1e-6)
        np.testing.assert_allclose(O.precision(Ld).numpy(), g[f"prec_K{K}"],
                                   rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(
            O.maha(mean.detach(), T(g[f"other_K{K}"]), Ld).numpy(),
            g[f"maha_K{K}"], rtol=1e-6)


def test_mlp_and_losses(golden):
    g = golden("mlp_losses")
    for act in ("tanh", "relu", "leaky_relu", "softplus"):
        params = [(T(g[f"{act}_p{2 * i}"]), T(g[f"{act}_p{2 * i + 1}"]))
                  for i in range(3)]
        y = O.mlp_forward(params, T(g[f"{act}_x"]), act)
        np.testing.assert_allclose(y.numpy(), g[f"{act}_y"], rtol=1e-6,
                                   atol=1e-7)
        # same init from the same seed (orthogonal, zero bias, gains)
        torch.manual_seed(int(g[f"{act}_seed"]))
        init = O.mlp_init(11, 1, [16, 16], 1.0)
        for i, (W, b) in enumerate(init):
            assert np.array_equal(W.numpy(), g[f"{act}_p{2 * i}"])
            assert np.array_equal(b.numpy(), g[f"{act}_p{2 * i + 1}"])
    v, r, o = T(g["vl_values"]), T(g["vl_returns"]), T(g["vl_old"])
    np.testing.assert_allclose(O.value_loss(v, r, o, 0.0), g["vl_unclipped"],
                               rtol=1e-6)
    np.testing.assert_allclose(O.value_loss(v, r, o, 0.2), g["vl_clipped"],
                               rtol=1e-6)
    sl, ratio = O.surrogate_loss(T(g["sl_adv"]), T(g["sl_new"]),
                                 T(g["sl_old"]))
    np.testing.assert_allclose(sl, g["sl_loss"], rtol=1e-6)
    np.testing.assert_allclose(ratio, g["sl_ratio"], rtol=1e-6)
    gs = [T(g["gn_g0"]).clone(), T(g["gn_g1"]).clone()]
    before, after = O.grad_norm_clip(0.5, gs)
    np.testing.assert_allclose(before, g["gn_before"], rtol=1e-6)
    np.testing.assert_allclose(after, g["gn_after"], rtol=1e-5)
    np.testing.assert_allclose(gs[0].numpy(), g["gn_c0"], rtol=1e-5)
    assert O.mlp_arch_3_params(128, 2, 0.0) == list(g["arch_128_2_0"])
    assert O.mlp_arch_3_params(64, 3, -0.5) == list(g["arch_64_3_m05"])
    assert O.mlp_arch_3_params(256, 1, 0.0) == list(g["arch_256_1_0"])


def test_rms(golden):
    g = golden("rms")
    rms = O.RunningMeanStd((6,))
    for i in range(3):
        rms.update(T(g[f"arr_{i}"]))
    np.testing.assert_allclose(rms.mean.numpy(), g["mean"], rtol=1e-6)
    np.testing.assert_allclose(rms.var.numpy(), g["var"], rtol=1e-6)
    np.testing.assert_allclose(rms.count, g["count"], rtol=1e-12)


def test_mdp_reward(golden):
    g = golden("mdp_reward")
    out = O.make_mdp_reward(T(g["r"]), T(g["flags"]))
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=1e-6, atol=1e-7)
    assert np.array_equal(g["noop"], g["r"])       # non-TT/Hopper task ids


def test_pair_logprob_plumbing(golden):
    """Gather order / dof-major flattening / pair expansion / MVN call of
    TemporalCorrelatedPolicy.log_prob (reference code run with this oracle's
    ProDMP injected) == the oracle's own pair_log_prob."""
    g = golden("pair_logprob_plumbing")
    for tag in ("mw", "bp"):
        cfg = {k[len(tag) + 5:]: g[k].item() for k in g.files
               if k.startswith(tag + "_cfg_")}
        mp = ProDMPOracle(dtype=torch.float32, **cfg)
        a = lambda k: T(g[f"{tag}_{k}"])
        pos, vel = mp.sample_trajectories(a("times"), a("mean"), a("L"),
                                          a("t0"), a("y0"), a("v0"), a("eps"))
        np.testing.assert_allclose(torch.cat([pos, vel], -1).numpy(),
                                   g[f"{tag}_traj"], rtol=1e-6, atol=1e-6)
        lp = pair_log_prob(mp, a("traj"), a("mean"), a("L"), a("times"),
                           a("t0"), a("y0"), a("v0"), a("pairs"))
        np.testing.assert_allclose(lp.numpy(), g[f"{tag}_logp"], rtol=1e-5,
                                   atol=1e-5)
