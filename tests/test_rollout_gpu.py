"""GPU parity of the rollout-buffer kernels (obs running mean/std, MDP reward)
against the golden vectors generated from the reference and the oracle."""
import numpy as np
import pytest
import torch

from oracle import tce_oracle as O

pytestmark = pytest.mark.gpu
T_ = torch.as_tensor


@pytest.fixture(scope="module")
def ops():
    from tce_rl_amd import ops
    return ops


def test_rms_golden(ops, golden):
    g = golden("rms")
    mean = torch.zeros(6, device="cuda")
    var = torch.ones(6, device="cuda")
    count = 1e-4
    for i in range(3):
        count = ops.rms_update(T_(g[f"arr_{i}"]).cuda(), mean, var, count)
    np.testing.assert_allclose(mean.cpu().numpy(), g["mean"], rtol=2e-6,
                               atol=1e-7)
    np.testing.assert_allclose(var.cpu().numpy(), g["var"], rtol=2e-6)
    np.testing.assert_allclose(count, g["count"], rtol=1e-12)


@pytest.mark.parametrize("R,D,dtype", [(2050, 48, torch.float32),
                                       (1001, 35, torch.float32),
                                       (4096 * 501, 48, torch.float32),
                                       (3000, 20, torch.float64)])
def test_rms_update_and_normalize_vs_oracle(ops, R, D, dtype):
    g = torch.Generator(device="cuda").manual_seed(R)
    x = torch.randn(R, D, device="cuda", generator=g, dtype=dtype) * 3 + 2
    rms = O.RunningMeanStd((D,), dtype)
    mean = torch.zeros(D, device="cuda", dtype=dtype)
    var = torch.ones(D, device="cuda", dtype=dtype)
    count = 1e-4
    for rep in range(2):
        xi = x + rep
        rms.update(xi.cpu())
        count = ops.rms_update(xi, mean, var, count)
    tol = 2e-5 if dtype == torch.float32 else 1e-10
    torch.testing.assert_close(mean.cpu(), rms.mean, rtol=tol, atol=tol)
    torch.testing.assert_close(var.cpu(), rms.var, rtol=tol, atol=tol)
    assert abs(count - rms.count) < 1e-6
    y = ops.rms_normalize(x[:1000], mean, var)
    torch.testing.assert_close(y.cpu(), rms.normalise(x[:1000].cpu()),
                               rtol=1e-5, atol=1e-5)


def test_mdp_reward_golden(ops, golden):
    g = golden("mdp_reward")
    out = ops.mdp_reward(T_(g["r"]).cuda(), T_(g["flags"]).cuda())
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=1e-6,
                               atol=1e-6)
    # idempotent on its own output when the event flags stay the same
    again = ops.mdp_reward(out, T_(g["flags"]).cuda())
    np.testing.assert_allclose(again.cpu().numpy(), out.cpu().numpy(),
                               rtol=1e-6, atol=1e-6)
