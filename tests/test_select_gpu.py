"""Radix-select median (csrc/select.hip, the "median" entries of the metric
dictionaries) against torch.median (a sort) -- bit for bit, both float types,
ties, signed zeros, infinities, sizes from 1 to a few million."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("n", [1, 2, 3, 255, 256, 4097, 100_000, 2_052_096])
def test_median_equals_the_sorted_one(n, dtype):
    from tce_rl_amd import ops
    g = torch.Generator(device="cuda").manual_seed(n)
    x = torch.randn(n, device="cuda", dtype=dtype, generator=g) * 37.0
    ref = x.median()
    got = ops.median(x)
    assert got.dtype == torch.float64 and got.dim() == 0
    assert got.item() == ref.double().item()
    # a second call reuses the workspace the first one left zeroed
    assert ops.median(-x).item() == (-x).median().double().item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_median_ties_zeros_infinities_and_shapes(dtype):
    from tce_rl_amd import ops
    cases = [
        torch.tensor([3.0, 3.0, 3.0, 3.0]),
        torch.tensor([0.0, -0.0, 0.0, -0.0, 1.0]),
        torch.tensor([float("inf"), -float("inf"), 1.0, -1.0, 0.5, 2.0]),
        torch.tensor([1e-40, -1e-40, 0.0]),                     # subnormals (fp32)
        torch.arange(10001, dtype=torch.float64).flip(0) - 5000,
        (torch.arange(64 * 333) % 7).double().reshape(64, 333),  # many ties, 2-D
    ]
    for c in cases:
        x = c.to(dtype).cuda()
        assert ops.median(x).item() == x.reshape(-1).median().double().item(), c[:8]
    # a strided view is made contiguous first
    y = torch.randn(300, 70, device="cuda", dtype=dtype)[:, ::2]
    assert ops.median(y).item() == y.reshape(-1).median().double().item()


def test_device_stats_use_it():
    from tce_rl_amd import util
    x = torch.randn(5000, device="cuda")
    st = util.device_stats({"x": x, "n": torch.arange(7, device="cuda")}, "t")
    assert st["t_x_median"] == x.median().double().item()
    assert st["t_n_median"] == 3.0                           # integers: the library path
    assert abs(st["t_x_mean"] - x.double().mean().item()) < 1e-12
