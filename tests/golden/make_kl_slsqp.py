"""SLSQP solutions of the constrained covariance projection
    min KL_cov(S~ || S)  s.t.  KL_cov(S~ || S_old) <= eps
for the inputs tests/test_gauss_gpu.py feeds the projection KERNEL
(tests/test_kl_optimum_cpu.py:direct_cov_projection; 3 restarts of scipy's
SLSQP on the Cholesky parameters).  At K = 24 (300 unknowns) one solve takes
minutes on a GPU box's host share, so the GPU test reads the solutions from
tests/golden/kl_slsqp.npz; tests/test_kl_optimum_cpu.py re-solves the small
cases and checks them against the file.

    python tests/golden/make_kl_slsqp.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))   # the repo root (oracle/)
from test_kl_optimum_cpu import direct_cov_projection, spd  # noqa: E402

EPS = 5e-3
out = {}
for K in (4, 12, 24):
    g = np.random.default_rng(K)
    S_old, S = spd(K, g), spd(K, g, scale=1.7)
    C, f, slack = direct_cov_projection(S, S_old, EPS)
    out["S_%d" % K], out["S_old_%d" % K] = S, S_old
    out["C_%d" % K] = C
    out["f_slack_%d" % K] = np.array([f, slack])
    print(K, f, slack)
out["eps"] = np.array(EPS)
np.savez_compressed(os.path.join(HERE, "kl_slsqp.npz"), **out)
