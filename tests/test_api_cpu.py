"""API mirrors of the reference that need no GPU (host logic only)."""
import numpy as np
import torch


def test_running_mean_std_moment_merge_equals_pooled_statistics():
    """RunningMeanStd.update_from_moments / combine / copy
    (mprl/util/util_numerical.py:296-337): merging the moments of two batches
    gives the pooled mean and (population-weighted) variance."""
    from tce_rl_amd.rl.sampler import RunningMeanStd
    g = torch.Generator().manual_seed(0)
    a = torch.randn(40, 5, generator=g, dtype=torch.float64) * 3 + 1
    b = torch.randn(70, 5, generator=g, dtype=torch.float64) - 2
    r = RunningMeanStd(shape=(5,), dtype="float64", device="cpu", epsilon=0.0)
    r.update_from_moments(a.mean(0), a.var(0, unbiased=False), a.shape[0])
    other = RunningMeanStd(shape=(5,), dtype="float64", device="cpu",
                           epsilon=0.0)
    other.update_from_moments(b.mean(0), b.var(0, unbiased=False), b.shape[0])
    snapshot = r.copy()
    r.combine(other)
    both = torch.cat([a, b])
    np.testing.assert_allclose(r.mean.numpy(), both.mean(0).numpy(), rtol=1e-12)
    np.testing.assert_allclose(r.var.numpy(),
                               both.var(0, unbiased=False).numpy(), rtol=1e-12)
    assert r.count == 110
    # the copy is detached from the original
    np.testing.assert_allclose(snapshot.mean.numpy(), a.mean(0).numpy(),
                               rtol=1e-12)
    assert snapshot.count == 40


def test_experiment_keeps_the_references_static_helpers():
    """mp_exp.py:105-162: finalize (cw2 hook), get_dim_in, dim_policy_out."""
    from tce_rl_amd.mp_exp import MPExperiment

    class S:
        observation_shape = (4, 31)
    cfg = {"mp": {"type": "prodmp", "args": {"num_dof": 7, "num_basis": 8}},
           "sampler": {"type": "TemporalCorrelatedSampler"}}
    assert MPExperiment.dim_policy_out(cfg) == 63
    assert MPExperiment.get_dim_in(cfg, S()) == 31 - 14
    cfg["sampler"]["type"] = "BlackBoxSampler"
    assert MPExperiment.get_dim_in(cfg, S()) == 31
    cfg["mp"]["args"]["disable_goal"] = True
    assert MPExperiment.dim_policy_out(cfg) == 56
    assert MPExperiment().finalize() is None
