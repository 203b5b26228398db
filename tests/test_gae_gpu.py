"""GPU parity: HIP GAE / segment-advantage kernels vs the golden vectors
(generated from the reference) and vs the CPU oracle on seeded inputs."""
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


def dev(x):
    return T_(x).cuda()


def test_gae_golden_bit_exact(ops, golden):
    g = golden("gae")
    for c in range(int(g["num_cases"])):
        adv, ret = ops.gae(dev(g[f"r_{c}"]), dev(g[f"v_{c}"]),
                           dev(g[f"dones_{c}"]), dev(g[f"tl_{c}"]),
                           float(g[f"gamma_{c}"]), 0.95,
                           bool(g[f"use_gae_{c}"]))
        assert np.array_equal(adv.cpu().numpy(), g[f"adv_{c}"]), c
        assert np.array_equal(ret.cpu().numpy(), g[f"ret_{c}"]), c


# (T % 4 != 0 with T >= 4 takes the 16-byte accesses with the straddling lane
# shifted -- round 6 --, T < 4 the element path, T > 512 several tiles; 350 =
# table tennis's horizon, 4097 envs x 350: several waves per SIMD = whole-tile
# prefetch; every residue of T mod 4 and every compile-time pass count)
@pytest.mark.parametrize("N,T", [(1, 1), (2, 2), (3, 3), (3, 4), (3, 5),
                                 (3, 6), (3, 7), (5, 65), (4, 127), (7, 130),
                                 (5, 258), (6, 322), (11, 350), (4, 387),
                                 (3, 449), (5, 511), (5, 513), (9, 1100),
                                 (9, 1101), (4, 1027), (130, 500),
                                 (4097, 350), (4100, 101)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_gae_vs_oracle_ragged(ops, N, T, dtype):
    g = torch.Generator().manual_seed(N * 1000 + T)
    r = torch.randn(N, T, generator=g, dtype=dtype)
    v = torch.randn(N, T + 1, generator=g, dtype=dtype)
    d = torch.rand(N, T, generator=g) < 0.01
    tl = torch.rand(N, T, generator=g) < 0.01
    for use_gae in (True, False):
        a0, r0 = O.gae(r, v, d, tl, 0.99, 0.95, use_gae)
        a1, r1 = ops.gae(r.cuda(), v.cuda(), d.cuda(), tl.cuda(), 0.99, 0.95,
                         use_gae)
        assert torch.equal(a1.cpu(), a0) and torch.equal(r1.cpu(), r0)


def test_segment_advantage_golden(ops, golden):
    g = golden("segment_advantage")
    for c in range(int(g["num_cases"])):
        out = ops.segment_advantage(
            str(g[f"mode_{c}"]), dev(g[f"r_{c}"]), dev(g[f"v_{c}"]),
            dev(g[f"a_{c}"]), dev(g[f"pairs_{c}"]), float(g[f"gamma_{c}"]),
            bool(g[f"norm_{c}"]), float(g[f"clip_{c}"]))
        # fp tolerance 1e-5 (north star); indexing is exact
        np.testing.assert_allclose(out.cpu().numpy(), g[f"out_{c}"],
                                   rtol=1e-5, atol=1e-5, err_msg=str(c))


def test_fused_gae_segadv_full_size(ops):
    """BASELINE config C2 shape (N 4096, T 500, P 24): fused path == oracle on
    a slice, plus size-independent properties on the whole batch."""
    N, T = 4096, 500
    g = torch.Generator().manual_seed(0)
    r = torch.randn(N, T, generator=g)
    v = torch.randn(N, T + 1, generator=g)
    d = torch.zeros(N, T, dtype=torch.bool)
    d[:, -1] = True
    d |= torch.rand(N, T, generator=g) < 0.002
    tl = torch.zeros_like(d)
    torch.manual_seed(0)
    pairs = O.get_time_pairs(T, dict(num_select=25, fixed_interval=True))
    adv, ret, seg, partials = ops.gae(r.cuda(), v.cuda(), d.cuda(), tl.cuda(),
                                      1.0, 0.95, True, pairs.cuda())
    # identity: adv == ret - V[:, :-1] exactly
    assert torch.equal(adv.cpu(), ret.cpu() - v[:, :-1])
    # terminal step: ret_t = (r_t - V_t) + V_t ~ r_t where done
    torch.testing.assert_close(ret.cpu()[d], r[d], rtol=1e-5, atol=1e-5)
    sl = slice(1000, 1064)
    a0, r0 = O.gae(r[sl], v[sl], d[sl], tl[sl], 1.0, 0.95, True)
    assert torch.equal(adv.cpu()[sl], a0) and torch.equal(ret.cpu()[sl], r0)
    out = ops.segment_advantage("value_subtraction", r.cuda(), v.cuda(), adv,
                                pairs.cuda(), 1.0, True, 0.0,
                                fused=(seg, partials)).cpu()
    ref = O.segment_advantage("value_subtraction", r, v, a0, pairs, 1.0, True)
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    # normalised output: mean 0, unbiased std 1
    assert abs(out.double().mean().item()) < 1e-5
    assert abs(out.double().std().item() - 1) < 1e-5
    # linearity of the raw segment advantage in (r, V)
    _, _, seg2, _ = ops.gae(2 * r.cuda(), 2 * v.cuda(), d.cuda(), tl.cuda(),
                            1.0, 0.95, True, pairs.cuda())
    assert torch.equal(seg2, 2 * seg)


def test_moments_normalize(ops):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(100003, generator=g) * 3 + 1
    st = ops.moments(x.cuda()).cpu()
    assert st[0].item() == x.numel()
    np.testing.assert_allclose(st[1].item(), x.double().mean().item(),
                               rtol=1e-12)
    np.testing.assert_allclose((st[2] / (st[0] - 1)).sqrt().item(),
                               x.double().std().item(), rtol=1e-10)
    y = ops.normalize(x.cuda(), ops.moments(x.cuda()), clip=2.0).cpu()
    ref = torch.clamp((x - x.mean()) / (x.std() + 1e-8), -2, 2)
    np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)
    # BBRL guard: a single element normalises with std := 1
    one = torch.tensor([3.0]).cuda()
    assert ops.normalize(one, ops.moments(one), single_std_one=True).item() == 0
