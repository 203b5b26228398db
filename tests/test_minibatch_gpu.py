"""Minibatched critic epochs on the matrix-core kernels.

The reference's critic update (mprl/rl/agent/temporal_correlated_agent.py:
343-366, black_box_agent.py:124-146; class default ``num_minibatchs = 10``)
permutes the flattened rows with numpy's global generator
(``generate_minibatches``, mprl/util/util_data_structure.py:378-391), cuts the
permutation with ``np.array_split`` and takes one clipped Adam step per piece
on the GATHERED rows (``select_batch``).  ``tce_mlp_critic_minibatch_f32`` /
``tce_mlpw_critic_minibatch_*`` do that as ONE C call per epoch, reading the
gathered rows in place.  Compared here against torch: the same pieces through
``F.linear`` autograd in float64, ``clip_grad_norm_``-style clipping and
``torch.optim.Adam`` -- parameters after the epoch, per-piece loss and
gradient norms."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ACTS = {"tanh": torch.tanh, "relu": F.relu, "leaky_relu": F.leaky_relu,
        "softplus": F.softplus}


def _net(din, hidden, act, dtype, seed):
    from tce_rl_amd.nn import MLP
    torch.manual_seed(seed)
    return MLP("ValueFunction", din, 1, [hidden, hidden], "orthogonal", 1.0,
               act, None, dtype, torch.device("cuda"))


def _reference_epoch(ws, opt, act, x, ret, old, clip, perm, k, grad_clip):
    """The reference's loop over the pieces, float64 on the device."""
    rows = []
    for piece in np.array_split(perm, k):
        sel = torch.as_tensor(piece, device=x.device)
        h = ACTS[act](F.linear(x[sel], ws[0], ws[1]))
        h = ACTS[act](F.linear(h, ws[2], ws[3]))
        v = F.linear(h, ws[4], ws[5]).squeeze(-1)
        r, o = ret[sel], old[sel]
        loss = (r - v).pow(2)
        if clip > 0:
            vc = o + (v - o).clamp(-clip, clip)
            loss = torch.max(loss, (vc - r).pow(2))
        loss = loss.mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        gn = torch.sqrt(sum((w.grad ** 2).sum() for w in ws))
        gc = gn
        if grad_clip > 0:
            # util.grad_norm_clip (mprl/util/util_numerical.py:244-275)
            torch.nn.utils.clip_grad_norm_(ws, grad_clip)
            gc = torch.sqrt(sum((w.grad ** 2).sum() for w in ws))
        opt.step()
        rows.append([loss.item(), gn.item(), gc.item()])
    return np.asarray(rows)


CASES = [
    # (hidden, dtype, din, N, T, k, act, clip_critic, grad_clip, weight_decay)
    (128, torch.float32, 40, 37, 50, 10, "relu", 0.0, 0.0, 0.0),
    (128, torch.float32, 22, 9, 64, 3, "tanh", 0.5, 0.0, 1e-3),
    (128, torch.float32, 39, 64, 31, 7, "leaky_relu", 0.0, 0.3, 0.0),
    (128, torch.float32, 17, 200, 1, 4, "softplus", 0.7, 0.5, 0.0),
    (256, torch.float32, 22, 40, 35, 10, "leaky_relu", 0.0, 0.0, 0.0),
    (256, torch.float32, 40, 11, 60, 4, "relu", 0.5, 0.4, 1e-3),
    (256, torch.float64, 22, 30, 33, 10, "leaky_relu", 0.0, 0.0, 0.0),
    (128, torch.float64, 40, 21, 40, 3, "tanh", 0.6, 0.5, 0.0),
]


@pytest.mark.parametrize(
    "hidden,dtype,din,N,T,k,act,clip,grad_clip,wd", CASES,
    ids=lambda v: str(v).replace("torch.", ""))
def test_minibatched_epochs_equal_torch(hidden, dtype, din, N, T, k, act, clip,
                                        grad_clip, wd):
    from tce_rl_amd import critic_ops
    from tce_rl_amd.optim import FlatAdam
    mlp = _net(din, hidden, act, dtype, 0)
    assert critic_ops.supported(mlp)
    g = torch.Generator(device="cuda").manual_seed(1)
    full = torch.randn(N, T + 1, din + 8, device="cuda", generator=g,
                       dtype=dtype)
    x = full[:, :-1, :din]                       # strided view, like the agent
    values = torch.randn(N, T + 1, device="cuda", generator=g, dtype=dtype)
    old = values[:, :-1]                         # strided too
    ret = torch.randn(N, T, device="cuda", generator=g, dtype=dtype) * 2
    R = N * T
    # the reference side (float64 copies of the same initial weights)
    ws = [p.detach().double().clone().requires_grad_(True)
          for p in mlp.parameters()]
    ref_opt = torch.optim.Adam(ws, lr=3e-3, weight_decay=wd)
    opt = FlatAdam(list(mlp.parameters()), lr=3e-3, weight_decay=wd)
    run = critic_ops.make_runner(mlp, opt.flat_grad)
    opt.bind_grads()
    E = 2
    rng = np.random.RandomState(7)
    rows = torch.zeros(E * k, 4, dtype=dtype, device="cuda")
    ref_rows = []
    for e in range(E):
        perm = np.arange(R)
        rng.shuffle(perm)
        ref_rows.append(_reference_epoch(
            ws, ref_opt, act, x.reshape(R, din).double(),
            ret.reshape(R).double(), old.reshape(R).double(), clip, perm, k,
            grad_clip))
        run.epoch_minibatches(x, ret, old, clip,
                              torch.as_tensor(perm, device="cuda"), k,
                              rows[e * k:(e + 1) * k], opt,
                              grad_clip=grad_clip)
    assert opt.host_step == E * k
    host = rows.double().cpu().numpy()
    ref_rows = np.concatenate(ref_rows)
    f32 = dtype == torch.float32
    rt = 2e-5 if f32 else 1e-10
    np.testing.assert_allclose(host[:, 0], ref_rows[:, 0], rtol=rt, atol=rt)
    if grad_clip > 0:
        np.testing.assert_allclose(host[:, 2], ref_rows[:, 1],
                                   rtol=50 * rt, atol=rt)
        np.testing.assert_allclose(host[:, 3], ref_rows[:, 2],
                                   rtol=50 * rt, atol=rt)
    else:
        np.testing.assert_allclose(np.sqrt(host[:, 1]), ref_rows[:, 1],
                                   rtol=50 * rt, atol=rt)
    # parameters after E * k Adam steps: Adam's step lr * m / (sqrt(v) + eps) is
    # scale free, so a RELATIVE gradient error delta moves a parameter by
    # ~ lr * delta per step -- and for the smallest gradient entries the
    # float32 rounding of a 2000-term sum is delta ~ 2e-3 of the entry
    # (float64: 1e-12): bound lr * steps * delta
    for p, w in zip(mlp.parameters(), ws):
        err = (p.detach().double() - w.detach()).abs().max().item()
        assert err <= 3e-3 * E * k * (2e-3 if f32 else 1e-9), (p.shape, err)


def test_one_piece_equals_the_full_batch_epoch():
    """k = 1 with the identity permutation is the un-gathered epoch, bit for
    bit (same rows in the same tiles, same summation order)."""
    from tce_rl_amd import critic_ops
    from tce_rl_amd.optim import FlatAdam
    out = []
    for mode in ("plain", "index"):
        mlp = _net(40, 128, "relu", torch.float32, 3)
        g = torch.Generator(device="cuda").manual_seed(5)
        x = torch.randn(300, 41, 48, device="cuda", generator=g)[:, :-1, :40]
        ret = torch.randn(300, 40, device="cuda", generator=g)
        opt = FlatAdam(list(mlp.parameters()), lr=1e-3)
        run = critic_ops.make_runner(mlp, opt.flat_grad)
        opt.bind_grads()
        rows = torch.zeros(1, 4, device="cuda")
        if mode == "plain":
            run.epoch(x, ret, ret, 0.0, stats=rows[0], adam=opt)
        else:
            run.epoch_minibatches(x, ret, ret, 0.0,
                                  torch.arange(300 * 40, device="cuda"), 1,
                                  rows, opt)
        out.append((rows.clone(), opt.flat_param.clone()))
    # loss and updated parameters bit for bit; |grad|^2 is an atomic sum over
    # the reduction's workgroups (order not fixed, last bits)
    assert torch.equal(out[0][0][:, 0], out[1][0][:, 0])
    torch.testing.assert_close(out[0][0][:, 1], out[1][0][:, 1], rtol=1e-6,
                               atol=0)
    assert torch.equal(out[0][1], out[1][1])


def test_minibatch_arguments_are_checked():
    from tce_rl_amd import critic_ops
    from tce_rl_amd.optim import FlatAdam
    mlp = _net(10, 128, "relu", torch.float32, 0)
    x = torch.randn(4, 5, 10, device="cuda")
    ret = torch.randn(4, 5, device="cuda")
    opt = FlatAdam(list(mlp.parameters()), lr=1e-3)
    run = critic_ops.make_runner(mlp, opt.flat_grad)
    rows = torch.zeros(30, 4, device="cuda")
    with pytest.raises(RuntimeError, match="more minibatches than rows"):
        run.epoch_minibatches(x, ret, ret, 0.0,
                              torch.arange(20, device="cuda"), 30, rows, opt)
    b3 = critic_ops.EpochRunner(mlp, opt.flat_grad, arith="bf16x3")
    with pytest.raises(NotImplementedError, match="exact-fp32"):
        b3.epoch_minibatches(x, ret, ret, 0.0,
                             torch.arange(20, device="cuda"), 2, rows[:2], opt)


@pytest.mark.parametrize("n", [1, 2, 3, 17, 1000, 4096, (1 << 20) + 7, 2048000])
def test_device_permutation_is_a_permutation(n):
    """``minibatch_permutation: device`` -- tce_feistel_permutation: a keyed
    bijection on [0, n) (Feistel network + cycle walking), every element on its
    own: each index exactly once, the same key the same permutation, another
    key another one, and rows spread over the pieces like a shuffle's."""
    from tce_rl_amd._lib import call, ptr, stream

    def draw(key):
        out = torch.empty(n, dtype=torch.int64, device="cuda")
        call("tce_feistel_permutation", ptr(out), n, key, stream())
        return out
    a = draw(12345)
    assert torch.equal(torch.sort(a).values, torch.arange(n, device="cuda"))
    assert torch.equal(a, draw(12345))
    if n >= 17:
        b = draw(12346)
        assert not torch.equal(a, b)
        assert torch.equal(torch.sort(b).values, torch.arange(n, device="cuda"))
    if n >= 4096:
        # no structure a minibatch would inherit: of the first tenth of the
        # rows about a tenth lands in each tenth of the permutation
        # (binomial: 5 sigma), few fixed points, neighbours are torn apart
        k = n // 10
        first = (a[:k * 10].reshape(10, k) < k).sum(1).double()
        sigma = (k * 0.1 * 0.9) ** 0.5
        assert float((first - 0.1 * k).abs().max()) <= 5 * sigma + 1
        assert int((a == torch.arange(n, device="cuda")).sum()) <= 8
        assert float(((a[1:] - a[:-1]).abs() == 1).double().mean()) < 1e-3


def test_device_permutation_positions_are_uniform_over_keys():
    """Where row 0 lands, over 4000 keys: uniform over the ten pieces
    (chi-square, 9 degrees of freedom, 99.9 % quantile 27.9)."""
    from tce_rl_amd._lib import call, ptr, stream
    n, keys = 1000, 4000
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    where = []
    for key in range(keys):
        call("tce_feistel_permutation", ptr(out), n, 7919 * key + 3, stream())
        where.append(out.argmin())            # the position that holds row 0
    pos = torch.stack(where).cpu().numpy()
    counts = np.bincount(pos * 10 // n, minlength=10)
    chi2 = float(((counts - keys / 10) ** 2 / (keys / 10)).sum())
    assert chi2 < 27.9, (chi2, counts)


def test_agent_with_device_drawn_minibatches_is_repeatable():
    """Two TCE agents with num_minibatchs = 4, minibatch_permutation = device,
    the same seeds (numpy's generator keys the permutations): identical
    parameters after two steps, on the fused minibatch epochs."""
    from tce_rl_amd.config import tce_config
    from tce_rl_amd.mp_exp import MPExperiment
    out = []
    for _ in range(2):
        cfg = tce_config("metaworld", num_env=32, num_basis=5, epochs=3,
                         evaluation_interval=0, seed=1)
        a = cfg["params"]["agent"]["args"]
        a["num_minibatchs"], a["minibatch_permutation"] = 4, "device"
        torch.manual_seed(4)
        np.random.seed(9)
        exp = MPExperiment()
        exp.initialize(cfg, 0, None)
        for _ in range(2):
            dict(exp.agent.step())
        assert exp.agent.last_critic_plan.kind == "fused-narrow"
        assert exp.agent.last_critic_plan.minibatches == 4
        out.append(torch.cat([p.detach().reshape(-1) for p in
                              exp.agent.critic.parameters]).clone())
    assert torch.equal(out[0], out[1])
