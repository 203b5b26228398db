"""GPU parity of the three-part bf16 critic kernel (csrc/mlpb.hip) against a
plain PyTorch fp64 reference of the same op.  Its operands are not narrower
than fp32 (x = b0 + b1 + b2 exactly), so it is held to what the exact-fp32
kernel is held to (tests/test_mlp_gpu.py) -- and, directly, to being no further
from the fp64 truth than that kernel on the same inputs."""
import pytest
import torch

from test_mlp_gpu import make, torch_ref

pytestmark = pytest.mark.gpu


ARITHS = ["bf16x3"]


def run_case(act, din, N, T, seed=0, scale_x=1.0, scale_ret=3.0, arith="bf16x3"):
    from tce_rl_amd import critic_ops
    mlp = make(din, act, seed)
    D = din + 8
    g = torch.Generator(device="cuda").manual_seed(seed + 1)
    full = torch.randn(N, T + 1, D, device="cuda", generator=g) * scale_x
    states = full[:, :-1]
    ret = torch.randn(N, T, device="cuda", generator=g) * scale_ret
    old = torch.randn(N, T, device="cuda", generator=g)
    x = states[..., :din]
    out = []
    for clip in (0.0, 0.7):
        v64, l64, g64 = torch_ref(mlp, x.reshape(-1, din), ret.reshape(-1),
                                  old.reshape(-1), clip, torch.float64)
        v32, l32, g32 = torch_ref(mlp, x.reshape(-1, din), ret.reshape(-1),
                                  old.reshape(-1), clip, torch.float32)
        ref = critic_ops.EpochRunner(mlp, arith="f32")
        ref.epoch(x, ret, old, clip)
        gf32 = [p.grad.clone() for p in mlp.parameters()]
        run = critic_ops.EpochRunner(mlp, arith=arith)
        stats = run.epoch(x, ret, old, clip).cpu()
        out.append((stats, l64, g64, g32, gf32,
                    [p.grad.clone() for p in mlp.parameters()]))
    return out


@pytest.mark.parametrize("act", ["relu", "tanh", "leaky_relu", "softplus"])
@pytest.mark.parametrize("din,N,T", [(40, 7, 33), (21, 5, 64), (32, 3, 1),
                                     (17, 130, 10), (31, 9, 21), (39, 4, 70),
                                     (33, 3, 40), (1, 6, 11)])
@pytest.mark.parametrize("arith", ARITHS)
def test_bf16x3_epoch_vs_torch(act, din, N, T, arith):
    for stats, l64, g64, g32, gf32, grads in run_case(act, din, N, T, arith=arith):
        assert abs(stats[0].item() - l64.item()) <= 1e-5 * abs(l64.item()) + 1e-6
        gn2 = sum((gg.double() ** 2).sum() for gg in g64).item()
        assert abs(stats[1].item() - gn2) <= 1e-4 * gn2 + 1e-9
        for gk, a, b in zip(grads, g64, g32):
            e = (gk.double() - a).abs().max().item()
            e32 = (b.double() - a).abs().max().item()
            scale = a.abs().max().item()
            assert e <= 4 * e32 + 1e-5 * scale + 1e-7, (gk.shape, e, e32, scale)
        # no further from the truth than the exact-fp32 kernel: the whole
        # gradient, relative to its norm (both sit in the fp32 rounding noise;
        # a factor 2 and 3e-7 apart at most)
        truth = torch.cat([t.reshape(-1) for t in g64])
        eb = (torch.cat([t.reshape(-1) for t in grads]).double() - truth).norm() / truth.norm()
        ea = (torch.cat([t.reshape(-1) for t in gf32]).double() - truth).norm() / truth.norm()
        assert eb.item() <= max(2 * ea.item(), 3e-7), (eb.item(), ea.item())


@pytest.mark.parametrize("scale_x,scale_ret", [(1e-3, 1e-3), (30.0, 500.0),
                                               (1.0, 1e-4), (1e6, 1e8),
                                               (1e-12, 1e-10)])
@pytest.mark.parametrize("arith", ARITHS)
def test_bf16x3_operand_ranges(scale_x, scale_ret, arith):
    """No range restriction: bf16 parts have the exponent range of fp32, so
    operands far outside the f16 range (where the split-f16 kernel reports
    inf) and tiny ones keep their 24 bits."""
    for stats, l64, g64, g32, gf32, grads in run_case("relu", 40, 33, 50, seed=3,
                                                      scale_x=scale_x,
                                                      scale_ret=scale_ret,
                                                      arith=arith):
        assert abs(stats[0].item() - l64.item()) <= 1e-5 * abs(l64.item()) + 1e-30
        for gk, a, b in zip(grads, g64, g32):
            e = (gk.double() - a).abs().max().item()
            e32 = (b.double() - a).abs().max().item()
            scale = a.abs().max().item()
            assert e <= 4 * e32 + 2e-5 * scale, (gk.shape, e, e32, scale)


@pytest.mark.parametrize("arith", ARITHS)
def test_bf16x3_c2_shape_is_as_close_to_fp64_as_the_fp32_kernel(arith):
    """BASELINE C2 rows (4096 x 500, D_in 40): relative error of the whole
    flat gradient against an fp64 PyTorch reference -- the three-part kernel
    may not be further away than the exact-fp32 kernel (x 1.25 for the noise
    of the comparison itself)."""
    from tce_rl_amd import critic_ops
    mlp = make(40, "relu", 5)
    g = torch.Generator(device="cuda").manual_seed(2)
    full = torch.randn(4096, 501, 48, device="cuda", generator=g)
    x = full[:, :-1, :40]
    ret = torch.randn(4096, 500, device="cuda", generator=g)
    _, l64, g64 = torch_ref(mlp, x.reshape(-1, 40), ret.reshape(-1),
                            ret.reshape(-1), 0.0, torch.float64)
    truth = torch.cat([t.reshape(-1) for t in g64])
    a = critic_ops.EpochRunner(mlp, arith="f32")
    sa = a.epoch(x, ret, ret, 0.0).cpu()
    ea = ((a.flat.double() - truth).norm() / truth.norm()).item()
    b = critic_ops.EpochRunner(mlp, arith=arith)
    sb = b.epoch(x, ret, ret, 0.0).cpu()
    eb = ((b.flat.double() - truth).norm() / truth.norm()).item()
    print("relative gradient error vs fp64: fp32 kernel %.3e, bf16x3 %.3e" % (ea, eb))
    assert eb <= 1.25 * ea + 1e-8
    assert abs(sb[0].item() - l64.item()) <= 2e-6 * abs(l64.item())
    assert abs(sa[0] - sb[0]).item() <= 2e-6 * abs(sa[0]).item()


@pytest.mark.parametrize("arith", ARITHS)
def test_bf16x3_values_output_and_workgroup_cap(arith):
    """The launch can also emit the values; a capped grid (the 224-workgroup
    epochs beside the policy stream) gives the same gradient up to the
    summation order of the slabs."""
    from tce_rl_amd import _lib, critic_ops
    from tce_rl_amd._lib import call, ptr, stream
    mlp = make(24, "tanh", 2)
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.randn(70000, 24, device="cuda", generator=g)
    ret = torch.randn(70000, device="cuda", generator=g)
    lib = _lib.load()
    P = lib.tce_mlp_critic_num_params(24)
    partials = torch.empty(lib.tce_mlp_critic_grid(), P + 2, device="cuda")
    grads = []
    for cap in (0, 224, 3):
        grad, stats = torch.empty(P, device="cuda"), torch.zeros(4, device="cuda")
        vals = torch.empty(70000, device="cuda")
        ws = [ptr(p) for p in mlp.parameters()]
        call("tce_mlp_critic_" + arith, ptr(x), 0, 24, 70000, 70000, 24, *ws, 0,
             ptr(ret), None, 0.0, ptr(vals), ptr(partials), ptr(grad), ptr(stats),
             cap, None, None, None, None, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, None,
             stream())
        ref = critic_ops.forward(mlp, x.expand(1, 70000, 24).contiguous())[0, :, 0]
        torch.testing.assert_close(vals, ref, rtol=1e-5, atol=1e-6)
        grads.append(grad)
    for gcap in grads[1:]:
        assert (gcap - grads[0]).abs().max().item() <= 1e-5 * grads[0].abs().max().item()
