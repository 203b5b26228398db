"""CPU oracle for the TCE/BBRL rollout + update hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / CPU baseline.  The
product path (``tce_rl_amd``) never imports this package and fails loudly when
its HIP library is missing.

Every function restates, on torch-CPU tensors and in the same operation order,
one piece of the reference's hot path and cites the reference file:line it
follows (paths relative to ``/root/reference/``).

Pinning status
--------------
* ``tce_oracle`` (pair selection, time grid, GAE, segment advantage, Cholesky
  head, param-space Gaussian, MLP, losses, running mean/std, mdp-reward): pinned
  against golden vectors produced by importing the reference's own functions in
  the build container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).
* ``prodmp_oracle`` (ProDMP basis / trajectory / covariance): **no reference
  vectors exist** -- the arithmetic lives in the un-vendored third-party package
  ``mp_pytorch==0.1.4`` (``conda_env.sh:41``), so parity with that package itself
  is unpinned.  It is restated from the ProDMP paper (Li et al., RA-L 2023,
  cited at ``README.md:221-233``) and the reference's call sites
  (``mprl/util/util_mp.py:11-46``,
  ``mprl/rl/policy/temporal_correlated_policy.py:74-92,188-192``) and pinned
  against an INDEPENDENT numerical solution: ``tests/prodmp_ode.py`` integrates
  the DMP ODE with scipy (DOP853), written from the paper and sharing no code
  with this package; ``tests/test_prodmp_ode_cpu.py`` bounds oracle trajectory,
  basis table and scale factors against it for every shipped MP configuration.
  The *index plumbing* of the pair-wise log-prob (gather order, dof-major
  flattening, MVN call) IS pinned by a golden generated from the reference's
  ``TemporalCorrelatedPolicy.log_prob`` with this oracle's ProDMP injected as
  ``policy.mp``.
* ``kl_oracle`` (KL trust-region projection): **no reference vectors exist**
  (third-party ``trust_region_projections`` @ ``TCE_ICLR24`` + ``cpp_projection``
  (ITPAL), ``conda_env.sh:34,56-60``: parity with those packages is unpinned);
  restated from Otto et al., ICLR 2021 and the reference's call sites
  (``mprl/rl/projection/__init__.py:18-40``,
  ``mprl/rl/agent/temporal_correlated_agent.py:439-441,530-567,641-686``) and
  pinned against an INDEPENDENT solution of the constrained problem it solves:
  ``tests/test_kl_optimum_cpu.py`` (scipy SLSQP on the Cholesky parameters), plus
  KKT / finite-difference self-tests.
* ``frob_oracle`` (Frobenius projection, named by the factory, selected by no
  experiment file): **parity unpinned** for the same reason; the paper's two
  closed forms sample by sample; ``tests/test_frob_gpu.py`` checks that both
  distances land on their bounds where they were exceeded.
"""
