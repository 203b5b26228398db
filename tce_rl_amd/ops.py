"""Torch-tensor front end of the HIP kernels (libtce_hip.so).

Every function takes HIP device tensors, enqueues kernels on the current
stream through the C ABI (include/tce_hip.h) and returns device tensors.
PyTorch is used for memory and streams only.  No CPU fallback.
"""
import torch

from . import _lib
from ._lib import call, ptr, stream, sfx, check_dev


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _bool_u8(t):
    """bool / uint8 tensor -> contiguous 1-byte view."""
    t = _c(t)
    if t.dtype == torch.bool:
        return t.view(torch.uint8)
    assert t.dtype == torch.uint8
    return t


# ---------------------------------------------------------------------------
# moments / normalisation
# ---------------------------------------------------------------------------
def merge_stats(stats, group=None):
    """All-gather (count, mean, M2) over the ranks of `group` and merge them
    (exact pooled-moments identity), so that a sharded batch normalises with
    the statistics of the global batch."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) \
            or dist.get_world_size(group) == 1:
        return stats
    w = dist.get_world_size(group)
    buf = stats.new_empty(w, 3)
    dist.all_gather_into_tensor(buf, stats.reshape(1, 3), group=group)
    n = buf[:, 0].sum()
    mean = (buf[:, 0] * buf[:, 1]).sum() / n
    m2 = (buf[:, 2] + buf[:, 0] * (buf[:, 1] - mean) ** 2).sum()
    return torch.stack([n, mean, m2])


def _finalize(partials, group=None):
    stats = torch.empty(3, dtype=torch.float64, device=partials.device)
    call("tce_moments_finalize", ptr(partials), partials.shape[0], ptr(stats),
         stream())
    return merge_stats(stats, group)


def moments(x, group=None):
    """{count, mean, M2} (float64[3], device) of all elements of x."""
    check_dev(x)
    x = _c(x)
    nparts = _lib.load().tce_moments_num_partials()
    partials = torch.empty(nparts, 3, dtype=torch.float64, device=x.device)
    call("tce_moments_partial_" + sfx(x.dtype), ptr(x), x.numel(),
         ptr(partials), stream())
    return _finalize(partials, group)


def normalize(x, stats=None, eps=1e-8, clip=0.0, single_std_one=False):
    """clamp((x - mean) / (std_unbiased + eps), +-clip)."""
    check_dev(x)
    x = _c(x)
    y = torch.empty_like(x)
    call("tce_normalize_" + sfx(x.dtype), ptr(x), ptr(y), x.numel(),
         ptr(stats), float(eps), float(clip), int(single_std_one), stream())
    return y


# ---------------------------------------------------------------------------
# GAE + segment advantage
# ---------------------------------------------------------------------------
def gae(rewards, values, dones, time_limit_dones, gamma, lam, use_gae=True,
        pred_pairs=None):
    """TemporalCorrelatedAgent.get_advantage_return on the GPU.

    Returns (adv, ret) or, with pred_pairs [P,2] int64, (adv, ret, seg_raw,
    partials): the un-normalised value_subtraction segment advantage [N,P] and
    the per-workgroup moment partials for its global normalisation."""
    check_dev(rewards, values, dones, time_limit_dones, pred_pairs)
    N, T = rewards.shape
    assert values.shape == (N, T + 1) and dones.shape == (N, T)
    assert values.dtype == rewards.dtype
    r, v = _c(rewards), _c(values)
    d, tl = _bool_u8(dones), _bool_u8(time_limit_dones)
    adv, ret = torch.empty_like(r), torch.empty_like(r)
    P, seg, partials, pairs = 0, None, None, None
    if pred_pairs is not None:
        pairs = _c(pred_pairs.to(torch.int64))
        P = pairs.shape[0]
        seg = torch.empty(N, P, dtype=r.dtype, device=r.device)
        partials = torch.empty(_lib.load().tce_gae_num_partials(N), 3,
                               dtype=torch.float64, device=r.device)
    call("tce_gae_" + sfx(r.dtype), ptr(r), ptr(v), ptr(d), ptr(tl), ptr(adv),
         ptr(ret), ptr(pairs), P, ptr(seg), ptr(partials), N, T, float(gamma),
         float(lam), int(bool(use_gae)), stream())
    if pred_pairs is None:
        return adv, ret
    return adv, ret, seg, partials


def segment_advantage(mode, rewards, values, advantages, pred_pairs, gamma,
                      norm_advantages=False, clip_advantages=0.0, group=None,
                      fused=None):
    """TemporalCorrelatedAgent.get_segment_advantage on the GPU.

    fused = (seg_raw, partials) from gae(..., pred_pairs) skips recomputing the
    value_subtraction sums."""
    check_dev(rewards, values, advantages, pred_pairs)
    N, T = rewards.shape
    pairs = _c(pred_pairs.to(torch.int64))
    P = pairs.shape[0]
    s = sfx(rewards.dtype)
    if mode == "value_subtraction":
        if fused is None:
            z = torch.zeros(N, T, dtype=torch.bool, device=rewards.device)
            _, _, seg, partials = gae(rewards, values, z, z, gamma, 0.0, True,
                                      pairs)
        else:
            seg, partials = fused
        if not norm_advantages:
            return seg
        return normalize(seg, _finalize(partials, group))
    if mode == "accumulate":
        adv = _c(advantages)
        stats = moments(adv, group) if norm_advantages else None
        out = torch.empty(N, P, dtype=adv.dtype, device=adv.device)
        call("tce_segment_accumulate_" + s, ptr(adv), ptr(pairs), P, ptr(out),
             N, T, ptr(stats), 1e-8, float(clip_advantages), stream())
        if norm_advantages:
            out = normalize(out, moments(out, group))
        return out
    if mode == "accumulated_rewards":
        r = _c(rewards)
        out = torch.empty(N, P, dtype=r.dtype, device=r.device)
        call("tce_segment_accrew_" + s, ptr(r), ptr(pairs), P, ptr(out), N, T,
             float(gamma), stream())
        return out
    raise NotImplementedError(mode)
