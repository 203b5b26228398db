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


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("n", [1, 2, 255, 4097, 2_048_000])
def test_stats5_matches_torch(dtype, n):
    """tce_stats5_*: the five entries of generate_stats
    (util_numerical.py:130-164) from one chain of launches == torch's
    reductions in float64 on the same tensor."""
    from tce_rl_amd import ops
    g = torch.Generator(device="cuda").manual_seed(n)
    x = torch.randn(n, device="cuda", generator=g, dtype=dtype) * 3 + 100
    got = ops.stats5(x).cpu()
    xd = x.double()
    want = torch.stack([xd.mean(), xd.max(), xd.min(), xd.median(),
                        xd.std() if n > 1 else xd.new_zeros(())]).cpu()
    assert got[1] == want[1] and got[2] == want[2] and got[3] == want[3]
    tol = 1e-12 if dtype == torch.float64 else 1e-9
    torch.testing.assert_close(got[0], want[0], rtol=tol, atol=0)
    # (the sum of squares about x[0]: ~1e-7 relative in the worst float32 case)
    torch.testing.assert_close(got[4], want[4], rtol=1e-6, atol=1e-12)
    # twice in a row on one stream: the workspace is left ready
    assert torch.equal(ops.stats5(x).cpu(), got)


def test_stats5_bool_int_and_nan():
    from tce_rl_amd import ops
    b = torch.zeros(1000, 7, dtype=torch.bool, device="cuda")
    b[:, -1] = True
    got = ops.stats5(b).cpu()
    bd = b.double()
    assert got[1:4].tolist() == [1.0, 0.0, 0.0]
    assert got[0].item() == pytest.approx(bd.mean().item(), rel=1e-14)
    torch.testing.assert_close(got[4], bd.std().cpu(), rtol=1e-9, atol=0)
    i = torch.arange(-5, 6, device="cuda")
    assert ops.stats5(i).cpu()[:4].tolist() == [0.0, 5.0, -5.0, 0.0]
    x = torch.randn(5000, device="cuda")
    x[17] = float("nan")
    got = ops.stats5(x).cpu()
    assert torch.isnan(got).all()                # as torch: every entry NaN
    # integers above float32's 2^24 keep their value ...
    big = torch.tensor([2 ** 24 + 1, 2 ** 24 + 3, 2 ** 24 + 5], device="cuda")
    assert ops.stats5(big).cpu()[:4].tolist() == [
        2.0 ** 24 + 3, 2.0 ** 24 + 5, 2.0 ** 24 + 1, 2.0 ** 24 + 3]
    # ... and an infinite FIRST element is not taken as the sums' shift:
    # mean / max as torch gives them (inf), not NaN
    y = torch.randn(4000, device="cuda", dtype=torch.float64)
    y[0] = float("inf")
    got = ops.stats5(y).cpu()
    assert got[0].item() == float("inf") and got[1].item() == float("inf")
    assert got[2].item() == y[1:].min().item()
    # an outlier first element: the variance does not cancel away
    z = torch.randn(100000, device="cuda", dtype=torch.float64)
    z[0] = 1e6
    torch.testing.assert_close(ops.stats5(z).cpu()[4], z.std().cpu(),
                               rtol=1e-9, atol=0)


def test_device_stats_uses_the_fused_chain():
    from tce_rl_amd import util
    # (odd counts: the host helper takes numpy's median -- the mean of the two
    # middle elements of an even count --, the device path torch's lower one)
    d = {"a": torch.randn(301, 21, device="cuda"),
         "flag": torch.rand(301, 21, device="cuda") > 0.5,
         "skip": "not a tensor"}
    out = util.device_stats(d, "pre")
    ref = util.generate_many_stats({k: v for k, v in d.items()
                                    if torch.is_tensor(v)}, "pre")
    assert set(out) == set(ref)
    for k in ref:
        assert out[k] == pytest.approx(ref[k], rel=1e-6, abs=1e-9), k
