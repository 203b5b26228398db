"""CPU restatement (torch-CPU) of the mprl-owned part of the hot path.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Pinned by the golden
vectors in ``tests/golden/`` (generated from the reference itself).
"""
import math

import numpy as np
import torch

LOG_2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------
# a1: segment boundaries            mprl/util/util_learning.py:74-150
# --------------------------------------------------------------------------
def select_pred_pairs(num_all, num_select=None, fixed_interval=False,
                      first_index=None):
    """Consecutive (idx[i], idx[i+1]) pairs as a float32 [P, 2] tensor.

    Draws from the *global* torch CPU generator exactly as the reference does
    (``torch.randint(low=0, high=interval+residual, size=[])`` for the fixed
    stride branch, ``torch.randperm`` otherwise; util_learning.py:107-122).
    """
    if num_select is None:
        assert not fixed_interval and first_index is None
        num_select = num_all
    assert num_select <= num_all
    if fixed_interval:
        interval = num_all // num_select
        residual = num_all % num_select
        if first_index is None:
            first_index = torch.randint(low=0, high=interval + residual,
                                        size=[]).item()
        assert 0 <= first_index < interval + residual
        idx = torch.arange(first_index, num_all, interval, dtype=torch.long)
    else:
        perm = torch.randperm(n=num_all)
        idx = torch.sort(perm[:num_select])[0]
    pairs = torch.zeros([idx.shape[0] - 1, 2])
    pairs[:, 0] = idx[:-1]
    pairs[:, 1] = idx[1:]
    return pairs


def get_time_pairs(num_times, time_pairs_config):
    """int64 pairs; temporal_correlated_sampler.py:80-85."""
    return select_pred_pairs(num_all=num_times,
                             **time_pairs_config).to(torch.long)


# --------------------------------------------------------------------------
# a2: time grid     temporal_correlated_sampler.py:64-78, util_matrix.py:139-192
# --------------------------------------------------------------------------
def get_times(init_time, dt, num_times):
    """times[n, i] = w_s[i]*(t0+dt) + w_e[i]*(t0+T*dt); [N, T]."""
    start = init_time + dt
    end = init_time + num_times * dt
    w_s = torch.linspace(1, 0, steps=num_times).to(start)
    w_e = torch.linspace(0, 1, steps=num_times).to(start)
    return w_s[None, :] * start[:, None] + w_e[None, :] * end[:, None]


# --------------------------------------------------------------------------
# a11: GAE          temporal_correlated_agent.py:118-181
# --------------------------------------------------------------------------
def gae(rewards, values, dones, time_limit_dones, gamma, lam, use_gae=True):
    """Reverse scan, same elementwise order as the reference."""
    gamma = torch.as_tensor(gamma, dtype=rewards.dtype)
    returns = torch.zeros_like(values)
    nd = torch.logical_not(dones)
    ntl = torch.logical_not(time_limit_dones)
    T = rewards.shape[1]
    disc = gamma * nd
    if use_gae:
        g = 0
        for t in reversed(range(T)):
            td = rewards[..., t] + disc[..., t] * values[..., t + 1] \
                 - values[..., t]
            g = td + disc[..., t] * lam * g
            g = g * ntl[..., t]
            returns[..., t] = g + values[..., t]
    else:
        returns[..., -1] = values[..., -1]
        for t in reversed(range(T)):
            returns[..., t] = ntl[..., t] * (
                    rewards[..., t] + disc[..., t] * returns[..., t + 1]) \
                              + time_limit_dones[..., t] * values[..., t]
    returns = returns[..., :-1]
    adv = returns - values[..., :-1]
    return adv.clone(), returns.clone()


# --------------------------------------------------------------------------
# a12: segment advantage      temporal_correlated_agent.py:183-321
# --------------------------------------------------------------------------
def _normalise(x):
    return (x - x.mean()) / (x.std() + 1e-8)      # unbiased std, eps outside


def segment_advantage(mode, rewards, values, advantages, pred_pairs, gamma,
                      norm_advantages=False, clip_advantages=0.0):
    dtype = rewards.dtype
    gamma = torch.as_tensor(gamma, dtype=dtype)
    if mode == "accumulate":
        adv = advantages
        if norm_advantages:
            adv = _normalise(adv)
        if clip_advantages > 0:
            adv = torch.clamp(adv, -clip_advantages, clip_advantages)
        out = torch.zeros(adv.shape[0], pred_pairs.shape[-2], dtype=dtype)
        for i, (lo, hi) in enumerate(pred_pairs):
            out[:, i] = torch.sum(adv[:, lo:hi + 1], dim=-1)   # inclusive end
        if norm_advantages:
            out = _normalise(out)
        return out
    start, end = pred_pairs[..., 0], pred_pairs[..., 1]
    idx = torch.arange(0, rewards.shape[-1], 1)
    disc_rewards = rewards * gamma.pow(idx)
    mask = torch.logical_and(start[:, None] <= idx,
                             idx < end[:, None]).to(dtype)
    acc = torch.einsum('ik,jk->ij', disc_rewards, mask)
    disc_first = gamma.pow(start)
    if mode == "value_subtraction":
        out = acc / disc_first + gamma.pow(end - start) * values[:, end] \
              - values[:, start]
        if norm_advantages:
            out = _normalise(out)
        return out
    if mode == "accumulated_rewards":
        return (acc - acc.mean(dim=0)) / disc_first
    raise NotImplementedError(mode)


# --------------------------------------------------------------------------
# a16: BBRL episode advantage        black_box_agent.py:90-103
# --------------------------------------------------------------------------
def bbrl_advantage(segment_reward, segment_value, norm_advantages,
                   clip_advantages):
    adv = segment_reward - segment_value
    if norm_advantages:
        std = adv.std() if len(adv) != 1 else 1.0
        adv = (adv - adv.mean()) / (std + 1e-8)
    if clip_advantages > 0:
        adv = torch.clamp(adv, -clip_advantages, clip_advantages)
    return adv


# --------------------------------------------------------------------------
# a3: Cholesky head   abstract_policy.py:166-197, util_matrix.py:12-55,
#                     util_numerical.py:44-95
# --------------------------------------------------------------------------
def to_softplus_space(x, lower_bound):
    lb = lower_bound if lower_bound is not None else 1e-2
    return torch.nn.functional.softplus(x) + lb


def reverse_from_softplus_space(x, lower_bound):
    lb = lower_bound if lower_bound is not None else 1e-2
    return torch.log(torch.exp(x - lb) - 1)


def build_lower_matrix(diag, off_diag):
    K = diag.shape[-1]
    L = diag.diag_embed()
    if off_diag is not None:
        row, col = torch.tril_indices(K, K, -1)
        L[..., row, col] = off_diag
    return L


def reverse_build_matrix(L, has_off_diag):
    diag = torch.diagonal(L, dim1=-2, dim2=-1)
    if not has_off_diag:
        return diag, None
    row, col = torch.tril_indices(L.shape[-1], L.shape[-1], -1)
    return diag, L[..., row, col]


def vector_to_cholesky(vec, dim_out, min_std, std_only):
    diag = to_softplus_space(vec[..., :dim_out], min_std)
    off = None if std_only else vec[..., dim_out:]
    return build_lower_matrix(diag, off)


def cholesky_to_vector(L, min_std, std_only):
    diag, off = reverse_build_matrix(L, not std_only)
    diag = reverse_from_softplus_space(diag, min_std)
    return diag if std_only else torch.cat([diag, off], dim=-1)


def initial_variance_vector(dim_out, std_only, dtype=torch.float32):
    """abstract_policy.py:111-116: softplus^-1(1) with the default 1e-2 bound."""
    n = dim_out if std_only else dim_out + dim_out * (dim_out - 1) // 2
    v = torch.zeros(n, dtype=dtype)
    v[:dim_out] += reverse_from_softplus_space(
        torch.ones(dim_out, dtype=dtype), None)
    return v


# --------------------------------------------------------------------------
# a6/a7: param-space Gaussian          black_box_policy.py:58-224
# --------------------------------------------------------------------------
def mvn_rsample(mean, L, eps):
    """loc + L @ eps (what MultivariateNormal.rsample does with its noise)."""
    return mean + torch.einsum('...ij,...j->...i', L, eps)


def mvn_log_prob(x, mean, L):
    return torch.distributions.MultivariateNormal(
        loc=mean, scale_tril=L, validate_args=False).log_prob(x)


def mvn_entropy(mean, L):
    return torch.distributions.MultivariateNormal(
        loc=mean, scale_tril=L, validate_args=False).entropy()


def covariance(L):
    return torch.einsum('...ij,...kj->...ik', L, L)


def log_determinant(L):
    return 2 * L.diagonal(dim1=-2, dim2=-1).log().sum(-1)


def precision(L):
    eye = torch.eye(L.shape[-1], dtype=L.dtype)
    return torch.cholesky_solve(eye, L, upper=False)


def maha(x, y, L):
    diff = (x - y)[..., None]
    return torch.linalg.solve_triangular(L, diff, upper=False) \
        .pow(2).sum([-2, -1])


# --------------------------------------------------------------------------
# a8: MLP / critic          util_nn.py:75-246, util_hyperparams.py:8-46
# --------------------------------------------------------------------------
def mlp_arch_3_params(avg_neuron, num_hidden, shape):
    assert avg_neuron >= 0 and -1.0 <= shape <= 1.0 and num_hidden >= 1
    shape = shape * avg_neuron
    arch = []
    for i in range(num_hidden):
        x = 2 * i / (num_hidden - 1) - 1 if num_hidden != 1 else 0.0
        d = int(np.floor(shape * x + avg_neuron))
        arch.append(1 if d == 0 else d)
    return arch


_ACT = {"tanh": torch.tanh, "relu": torch.nn.functional.relu,
        "leaky_relu": torch.nn.functional.leaky_relu,
        "softplus": torch.nn.functional.softplus}


def mlp_init(dim_in, dim_out, hidden_layers, out_layer_gain,
             dtype=torch.float32):
    """Orthogonal init, gain sqrt(2) hidden / out_layer_gain last, zero bias
    (util_nn.py:65-69,142,150,157-158).  Returns [(W, b), ...] with W [out,in]
    drawing from the global torch RNG in layer order like the reference."""
    dims = [dim_in] + list(hidden_layers) + [dim_out]
    params = []
    for i in range(len(dims) - 1):
        lin = torch.nn.Linear(dims[i], dims[i + 1], dtype=dtype)
        gain = out_layer_gain if i == len(dims) - 2 else 2 ** 0.5
        torch.nn.init.orthogonal_(lin.weight.data, gain=gain)
        lin.bias.data.zero_()
        params.append((lin.weight.data.clone(), lin.bias.data.clone()))
    return params


def mlp_forward(params, x, act_hidden, act_last=None):
    for W, b in params[:-1]:
        x = _ACT[act_hidden](torch.nn.functional.linear(x, W, b))
    W, b = params[-1]
    x = torch.nn.functional.linear(x, W, b)
    return _ACT[act_last](x) if act_last is not None else x


# --------------------------------------------------------------------------
# a13/a14 losses           temporal_correlated_agent.py:688-745
# --------------------------------------------------------------------------
def value_loss(values, returns, old_vs, clip_critic=0.0):
    loss = (returns - values).pow(2)
    if clip_critic > 0:
        vc = old_vs + (values - old_vs).clamp(-clip_critic, clip_critic)
        loss = torch.max(loss, (vc - returns).pow(2))
    return loss.mean()


def surrogate_loss(advantages, log_prob_new, log_prob_old):
    ratio = (log_prob_new - log_prob_old).exp()
    return -(ratio * advantages).mean(), ratio.mean()


def grad_norm_clip(bound, grads):
    """util_numerical.py:244-275 on a list of gradient tensors (in place)."""
    before = sum(g.norm(2).item() ** 2 for g in grads) ** 0.5
    if bound > 0:
        coef = min(bound / (before + 1e-6), 1.0)
        for g in grads:
            g.mul_(coef)
    after = sum(g.norm(2).item() ** 2 for g in grads) ** 0.5
    return before, after


# --------------------------------------------------------------------------
# a9: running mean / std        util_numerical.py:278-337
# --------------------------------------------------------------------------
class RunningMeanStd:
    def __init__(self, shape, dtype=torch.float32, epsilon=1e-4):
        self.mean = torch.zeros(shape, dtype=dtype)
        self.var = torch.ones(shape, dtype=dtype)
        self.count = epsilon

    def update(self, arr):
        self.update_from_moments(torch.mean(arr, dim=0),
                                 torch.var(arr, dim=0), arr.shape[0])

    def update_from_moments(self, b_mean, b_var, b_count):
        delta = b_mean - self.mean
        tot = self.count + b_count
        new_mean = self.mean + delta * b_count / tot
        m2 = self.var * self.count + b_var * b_count + \
             torch.square(delta) * self.count * b_count / (self.count + b_count)
        self.mean, self.var, self.count = new_mean, m2 / tot, tot

    def normalise(self, raw):
        """temporal_correlated_sampler.py:87-89 (eps inside the sqrt)."""
        return (raw - self.mean) / torch.sqrt(self.var + 1e-8)


# --------------------------------------------------------------------------
# a10: non-MDP -> MDP reward       util_experiment.py:261-328
# --------------------------------------------------------------------------
def make_mdp_reward(step_rewards, event_flags):
    """event_flags [N, T] bool: from the first True index on, sum the rewards,
    write the sum at that index and zero what follows.  No-op for rows whose
    first event index is 0 (never happened *or* happened at step 0: the
    reference tests ``event_index_first > 0``)."""
    r = step_rewards.clone()
    ev = torch.where(event_flags, 1.0, 0.0).to(r.dtype)
    first = torch.argmax(ev, dim=-1)
    after = (r * ev).sum(-1)
    happened = first > 0
    r[happened, first[happened]] = after[happened]
    later = torch.arange(r.size(1)).unsqueeze(0) > first.unsqueeze(1)
    r[torch.logical_and(happened.unsqueeze(1), later)] = 0
    return r


# --------------------------------------------------------------------------
# Adam with L2-in-gradient weight decay (torch.optim.Adam defaults;
# abstract_agent.py:76-81) and LinearLR(1 -> 0.01)  (:91-103)
# --------------------------------------------------------------------------
def adam_step(p, g, m, v, step, lr, wd=0.0, b1=0.9, b2=0.999, eps=1e-8):
    if wd != 0:
        g = g + wd * p
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def linear_lr(base_lr, it, total_iters, start=1.0, end=0.01):
    f = start + (end - start) * min(it, total_iters) / total_iters
    return base_lr * f
