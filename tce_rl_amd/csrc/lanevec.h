// Triangular solves with ONE vector spread over the lanes of a wave: lane k
// holds element k (K <= 64), the matrix sits in LDS.  Step r broadcasts
// element r (v_readlane) and every later lane takes its multiple of column r
// off: one fused multiply-add per step for the whole vector, instead of the
// K^2 / 2 dependent LDS round trips of a thread that owns the whole vector
// (the thread-per-env solves of round 1: 120 cycles per inner step with one
// wave per CU).  The loads of a vector are coalesced as a by-product.
//
// Used by the per-env pieces of the policy epoch: mean projection
// (mprl/rl/projection kl_projection.py `mean_projection`), the
// Mahalanobis parts of the KL diagnostics / trust-region loss
// (mprl/util/util_learning.py `gaussian_kl` callers in
// temporal_correlated_agent.py:523-612) and the MVN log-prob of a shared
// covariance (temporal_correlated_policy.py:188-192).
#pragma once
#include "common.h"

namespace {

__device__ inline float lv_bcast(float v, int r) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), r));
}
__device__ inline double lv_bcast(double v, int r) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), r);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), r);
  return __hiloint2double(hi, lo);
}

// NS systems at once: v[s] <- L[s]^-1 v[s].  L[s]: LDS image [K][KP] of a lower
// triangular matrix, rd[s] = 1 / L[s]_kk in lane k (0 in lanes >= K, whose v
// must be 0).  Column r + 1 is fetched while column r is applied.
template <typename real, int NS>
__device__ inline void lv_solve_lower(real (&v)[NS], const real* const (&L)[NS],
                                      const real (&rd)[NS], int K, int KP, int lane) {
  const int row = (lane < K ? lane : K - 1) * KP;
  real col[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) col[s] = L[s][row];
  for (int r = 0; r < K; ++r) {
    real nxt[NS];
    const int rn = r + 1 < K ? r + 1 : r;
#pragma unroll
    for (int s = 0; s < NS; ++s) nxt[s] = L[s][row + rn];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const real z = lv_bcast(v[s] * rd[s], r);
      if (lane > r) v[s] -= col[s] * z;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) col[s] = nxt[s];
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) v[s] *= rd[s];
}

// v[s] <- L[s]^-T v[s] (rows of L from the last to the first)
template <typename real, int NS>
__device__ inline void lv_solve_lower_t(real (&v)[NS], const real* const (&L)[NS],
                                        const real (&rd)[NS], int K, int KP, int lane) {
  const int c = lane < K ? lane : K - 1;
  real rowv[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) rowv[s] = L[s][(K - 1) * KP + c];
  for (int r = K - 1; r >= 0; --r) {
    real nxt[NS];
    const int rn = r > 0 ? r - 1 : 0;
#pragma unroll
    for (int s = 0; s < NS; ++s) nxt[s] = L[s][rn * KP + c];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const real q = lv_bcast(v[s] * rd[s], r);
      if (lane < r) v[s] -= rowv[s] * q;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) rowv[s] = nxt[s];
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) v[s] *= rd[s];
}

}  // namespace
