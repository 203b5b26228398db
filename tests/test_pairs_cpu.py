"""a1: the PRODUCT's pair selection (host integer logic,
tce_rl_amd/util select_pred_pairs as TemporalCorrelatedSampler.get_time_pairs
calls it) against the fixture produced by the reference's own
select_pred_pairs (mprl/util/util_learning.py:74-150) -- bit-exact indices AND
the position of the global torch generator afterwards."""
import numpy as np
import torch

from tce_rl_amd.util import select_pred_pairs


def _pairs(num_all, **cfg):
    # the float32 -> int64 cast of temporal_correlated_sampler.py:81-83
    p = select_pred_pairs(num_all=num_all, **cfg)
    assert p.dtype == torch.float32
    return p.to(torch.long).numpy()


def test_product_pair_selection_bit_exact(golden):
    g = golden("pred_pairs")
    n = 0
    for num_all in (100, 250, 350, 500):
        for s in range(10):
            torch.manual_seed(s)
            pairs = _pairs(num_all, num_select=25, fixed_interval=True)
            ref = g[f"fixed_T{num_all}_s{s}"]
            assert pairs.dtype == ref.dtype and pairs.shape == ref.shape
            assert np.array_equal(pairs, ref)
            nxt = torch.randint(0, 1 << 30, size=[]).item()
            assert nxt == int(g[f"fixed_T{num_all}_s{s}_next"])
            n += 1
    for s in range(3):
        torch.manual_seed(s)
        assert np.array_equal(
            _pairs(100, num_select=25, fixed_interval=False),
            g[f"random_T100_s{s}"])
        n += 1
    assert n == 43


def test_fixed_first_index_and_shapes():
    p = _pairs(500, num_select=25, fixed_interval=True, first_index=7)
    assert p.shape == (24, 2)
    assert p[0, 0] == 7 and np.all(p[:, 1] - p[:, 0] == 20)
    assert np.array_equal(p[1:, 0], p[:-1, 1])
    p = _pairs(350, num_select=25, fixed_interval=True, first_index=0)
    assert np.all(p[:, 1] - p[:, 0] == 14) and p[-1, 1] == 336
