"""Host-side helpers mirroring the parts of ``mprl.util`` the hot path uses.

Integer / RNG logic (pair selection) is reproduced exactly on the host; tensor
math goes through ``tce_rl_amd.ops`` (HIP kernels).
"""
import collections.abc
import random
import time

import numpy as np
import torch


# --- util_data_structure.py:59-78 -------------------------------------------
def parse_dtype_device(dtype, device):
    if isinstance(dtype, torch.dtype):
        target = dtype
    elif dtype in ("float32", "torch.float32"):
        target = torch.float32
    elif dtype in ("float64", "torch.float64"):
        target = torch.float64
    else:
        raise NotImplementedError
    return target, torch.device(device)


def to_np(x, dtype=None):
    if torch.is_tensor(x):
        return np.asarray(x.detach().cpu().numpy(), dtype)
    return np.asarray(x, dtype)


def to_ts(data, dtype=torch.float32, device="cpu"):
    if isinstance(data, list) and all(isinstance(d, np.ndarray) for d in data):
        data = np.asarray(data)
    return torch.as_tensor(data, dtype=dtype, device=device)


def assert_shape(data, shape):
    assert list(data.shape) == list(shape), (list(data.shape), list(shape))


# --- util_hyperparams.py:8-46 ------------------------------------------------
def mlp_arch_3_params(avg_neuron, num_hidden, shape):
    assert avg_neuron >= 0
    assert -1.0 <= shape <= 1.0
    assert num_hidden >= 1
    shape = shape * avg_neuron
    arch = []
    for i in range(num_hidden):
        x = 2 * i / (num_hidden - 1) - 1 if num_hidden != 1 else 0.0
        d = int(np.floor(shape * x + avg_neuron))
        arch.append(1 if d == 0 else d)
    return arch


# --- util_learning.py:74-150 ---------------------------------------------------
def select_pred_pairs(num_all, num_select=None, fixed_interval=False,
                      first_index=None, **unused):
    """Segment boundaries as a float32 [P, 2] host tensor.  Integer logic and
    the draw from the *global torch CPU generator* are the reference's, so the
    pairs (and the generator position afterwards) are bit-identical for the
    same seed."""
    if num_select is None:
        assert fixed_interval is False and first_index is None
        num_select = num_all
    else:
        assert num_select <= num_all
    if fixed_interval:
        interval = num_all // num_select
        residual = num_all % num_select
        if first_index is None:
            first_index = torch.randint(low=0, high=interval + residual,
                                        size=[]).item()
        else:
            assert 0 <= first_index < interval + residual
        idx = torch.arange(start=first_index, end=num_all, step=interval,
                           dtype=torch.long)
    else:
        perm = torch.randperm(n=num_all)
        idx = torch.sort(perm[:num_select])[0]
    pairs = torch.zeros([idx.shape[0] - 1, 2])
    pairs[:, 0] = idx[:-1]
    pairs[:, 1] = idx[1:]
    return pairs


# --- util_debug.py:140-152, 48-101 ------------------------------------------------
def set_global_random_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


_timers = {}


def run_time_test(lock, key="default", sync=True):
    """Stop-watch like the reference's, but device-synchronised so that GPU
    phases are timed for real (the reference's timer is launch-time only)."""
    if sync and torch.cuda.is_available():
        torch.cuda.synchronize()
    if lock:
        _timers[key] = time.perf_counter()
        return None
    return time.perf_counter() - _timers.pop(key)


# --- util_numerical.py:130-240 (metrics) ----------------------------------------------
def generate_stats(data, name=None, to_np_=False):
    """mean / max / min / median / std of a list, array or tensor."""
    if torch.is_tensor(data):
        data = data.detach().double().cpu().numpy()
    data = np.asarray(data, dtype=np.float64)
    prefix = name + "_" if name else ""
    if data.size == 0:
        return {}
    return {prefix + "mean": float(data.mean()),
            prefix + "max": float(data.max()),
            prefix + "min": float(data.min()),
            prefix + "median": float(np.median(data)),
            prefix + "std": float(data.std(ddof=1)) if data.size > 1 else 0.0}


def generate_many_stats(data_dict, name=None, to_np_=False):
    prefix = name + "_" if name else ""
    out = {}
    for k, v in data_dict.items():
        out.update(generate_stats(v, prefix + k))
    return out


def device_stats(tensors, name):
    """Same five statistics for a dict of device tensors with ONE host copy:
    the reductions run on the device, the scalars are gathered in one tensor."""
    return device_stats_async(tensors, name)()


def device_stats_async(tensors, name):
    """Enqueue the reductions now (current stream); the returned callable does
    the single host copy and builds the dict -- call it when the host would
    wait anyway."""
    keys = [k for k, v in tensors.items()
            if torch.is_tensor(v) and v.numel() > 0]
    if not keys:
        return dict
    if all(tensors[k].is_cuda for k in keys):
        # one [keys, 5] buffer, one chain of launches per tensor
        # (tce_stats5_*: the radix select's first pass carries mean / max /
        # min / std; no library reduction, no sort)
        from .. import ops
        stacked = torch.empty(len(keys), 5, dtype=torch.float64,
                              device=tensors[keys[0]].device)
        for i, k in enumerate(keys):
            ops.stats5(tensors[k], out=stacked[i])
    else:
        rows = []
        for k in keys:
            x = tensors[k].detach().double().reshape(-1)
            std = x.std() if x.numel() > 1 else x.new_zeros(())
            rows.append(torch.stack([x.mean(), x.max(), x.min(), x.median(),
                                     std]))
        stacked = torch.stack(rows)

    def finish():
        host = stacked.cpu().numpy()
        out = {}
        prefix = name + "_" if name else ""
        for k, r in zip(keys, host):
            for s, val in zip(("mean", "max", "min", "median", "std"), r):
                out[prefix + k + "_" + s] = float(val)
        return out
    return finish


class LazyMetrics(collections.abc.MutableMapping):
    """The metrics of an iteration whose device work may still be running: a
    mapping that fills itself (one blocking read of the device-side records) on
    first access.  `agent.step()` returns it so that the host can enqueue the
    next rollout behind the critic epochs instead of waiting for them and then
    leaving the GPU idle while it prepares ~50 small launches.

    Deliberately NOT a dict subclass: C-level consumers of a dict (the json
    encoder, ``dict(**m)`` fast paths) look at the underlying storage and would
    see an empty dict while the read is pending.  ``resolve()`` returns the
    plain dict (what the reference's ``step()`` returns); ``json.dumps`` of the
    mapping itself raises TypeError instead of printing "{}"; pickling resolves
    and stores the plain dict.  The read also runs the checks the reference
    runs inside the update (NaN losses): if the resolver raises, it stays in
    place and every later access raises again."""

    def __init__(self, resolver):
        self._resolver = resolver
        self._data = {}
        self._extra = {}                # set while pending: on top of the read values

    def resolve(self):
        r = self._resolver
        if r is not None:
            data = dict(r())            # may raise: then the resolver is kept
            data.update(self._extra)
            self._data, self._extra, self._resolver = data, {}, None
        return self._data

    @property
    def pending(self):
        return self._resolver is not None

    def __getitem__(self, k):
        return self.resolve()[k]

    def __setitem__(self, k, v):
        (self._extra if self.pending else self._data)[k] = v

    def __delitem__(self, k):
        del self.resolve()[k]

    def __iter__(self):
        return iter(self.resolve())

    def __len__(self):
        return len(self.resolve())

    def __contains__(self, k):
        return k in self.resolve()

    def __repr__(self):
        return repr(self.resolve())

    def __eq__(self, other):
        if isinstance(other, LazyMetrics):
            other = other.resolve()
        return self.resolve() == other

    def __or__(self, other):
        return {**self.resolve(), **other}

    def __ror__(self, other):
        return {**other, **self.resolve()}

    def copy(self):
        return dict(self.resolve())

    def __reduce__(self):
        return (dict, (self.resolve(),))


def rewrite_dict(d, prefix):
    return {prefix + "_" + k: v for k, v in d.items()}


# --- util_file.py:280-317 (checkpoint naming) -----------------------------------------
def get_nn_save_paths(log_dir, nn_name, epoch):
    import os
    s_path = os.path.join(log_dir, nn_name + "_parameters.pkl")
    w_path = os.path.join(log_dir, nn_name + "_weights")
    if epoch is not None:
        w_path = w_path + "_{:d}".format(epoch)
    return s_path, w_path


def get_training_state_save_path(log_dir, name, epoch):
    import os
    o_path = os.path.join(log_dir, name + "_state")
    if epoch is not None:
        o_path = o_path + "_{:d}".format(epoch)
    return o_path
