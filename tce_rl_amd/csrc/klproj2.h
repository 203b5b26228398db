// KL covariance projection WITHOUT an eigen-decomposition, on the float64
// matrix instruction (v_mfma_f64_16x16x4_f64), one workgroup (4 waves) per
// matrix, K <= 64.  Same contract as the Jacobi kernels of gauss.hip
// (kl_cov_proj_fwd/bwd_kernel), which stay available (tce_kl_proj_impl(0)).
//
// Math (Otto et al. 2021; call sites mprl/rl/agent/temporal_correlated_agent.py:
// 530-567 through mprl/rl/projection/__init__.py:18-40).  With the whitened
// factor A = Lo^-1 L (lower triangular), M = A A^T:
//   projected (whitened) covariance  Mt(eta) = (eta + 1) M W,  W = (eta M + I)^-1
//   KL(Sigma~ || Sigma_old)          h(eta) = 1/2 [tr(Mt - I) - logdet Mt]
//   eta: the root of h(eta) = eps (h convex, decreasing);  L~ = Lo chol(Mt).
// Every quantity is formed so that only SMALL numbers are added up (the Jacobi
// version got that from the eigenvalues):  with N' = (eta M + I) / (eta + 1) and
// W' = N'^-1 = (eta + 1) W
//   tr(Mt - I) = <M - I, W'> / (eta + 1),   logdet Mt = logdet M - logdet N',
//   logdet M = 2 sum log a_ii,  logdet N' = sum log (pivots of N', all near 1)
//   dh/deta = 1/2 [ <I - M, W - W^2> / eta - <I - M, W> / (eta + 1) ].
// Root: Newton on phi(eta) = h^-1/2 - eps^-1/2 (h ~ c / eta^2 for large eta, so
// phi is nearly linear), bracketed; started from the previous call's eta when
// the context is warm: 2 - 4 evaluations.  One evaluation = one in-place
// Gauss-Jordan inversion of N' (registers, one barrier per pivot) + W'^2.
// Backward: the implicit-function gradient in matrix form,
//   Mbar = (eta+1) W Sbar W - kappa (eta+1)/2 (W^2 - M^-1 W / (eta+1)),
//   kappa = <Sbar, D> / h_eta,  D = (I - M)(W - W^2) / eta,
//   Abar = 2 Mbar A = 2/(eta+1) [ (W' Sbar W') A - kappa/2 W' (W' A - A^-T) ],
//   Lbar = tril(Lo^-T Abar),
// Sbar = the Cholesky backward of the gradient w.r.t. C~ = chol(Mt).
// All of it was checked against the eigenvalue formulation of oracle/kl_oracle.py
// (forward 1e-14, backward 1e-12 relative) before it became a kernel.
//
// Context (double, per matrix): To [K*K] (Lo^-1) | A [K*K] | W' [K*K] | C~ [K*K] |
// tail {eta, active, alpha, kl0, To valid}.  warm_start != 0: the context was
// written by an earlier call for a nearby L AND THE SAME Lo (the policy update:
// Lo is the old policy's factor for all epochs) -- To and eta are reused.
//
// LDS: four 64 x 66 float64 images (pitch 66 = 2 mod 32: the operand reads of
// both product forms are bank-conflict free); everything is padded to 64 x 64
// with identity (triangular / SPD matrices) or zeros.
#pragma once
#include "common.h"
#include "mfma16.h"

// K <= 64 (klp2) and K <= 32 (klp2s: half the block steps of every inversion /
// factorisation and a quarter of their work, two waves -- the C2 / C5 sizes K 24 /
// 28, where the projection's forward is the longest link of a policy epoch)
#define KLP_NS klp2
#define KLP_N 64
#include "klproj2_impl.h"
#undef KLP_NS
#undef KLP_N
#define KLP_NS klp2s
#define KLP_N 32
#include "klproj2_impl.h"
#undef KLP_NS
#undef KLP_N
