// ProDMP basis evaluation shared by the trajectory and pair log-prob kernels.
//
// The arithmetic restates mp_pytorch==0.1.4 (third-party, absent from the
// reference tree; call sites mprl/util/util_mp.py:11-46 and
// mprl/rl/policy/temporal_correlated_policy.py:74-92,188-192) from the ProDMP
// paper; see oracle/prodmp_oracle.py for the CPU restatement it is tested
// against.
//
// Device table `tab` [M, 4 + 2*nbg] (row-major), pre-computed on the host on a
// grid of step dt/tau in scaled time over 5 tau:
//   col 0..3 : y1, y2, dy1, dy2 (homogeneous solutions and derivatives)
//   col 4..  : position basis (weights..., goal), already multiplied by the
//              per-column weight/goal scale; then the velocity basis likewise
// A query at time t interpolates two table rows linearly (same index clipping
// and the same two-branch lerp as torch.lerp).
#pragma once
#include "common.h"

#define TCE_MAXB 16   // num_basis + 1 <= 16

template <typename real>
struct MPParams {
  const real* tab;
  int M;            // table rows
  int nbg;          // num_basis + 1
  real tau, delay;
  real scaled_dt;   // dt / tau
  real inv_scale_g; // 1 / (scale of the goal column): unscaled goal basis
  int rel_goal;     // relative_goal
};

// Basis row of one (t, t0): registers
//   row[0] = c_y0->pos, row[1] = c_v0->pos, row[2] = c_y0->vel, row[3] = c_v0->vel
//   row[4 + b]            = H_pos[b]
//   row[4 + TCE_MAXB + b] = H_vel[b]          (already divided by tau)
// so that  pos_d = row0*y0_d + row1*v0_d + sum_b Hpos[b]*theta[d,b]  (same for vel).
#define TCE_ROWLEN (4 + 2 * TCE_MAXB)

template <typename real>
__device__ inline void mp_lerp_setup(const MPParams<real>& mp, real t, int& i0, real& w) {
  real s = (t - mp.delay) / mp.tau;
  s = s > real(0) ? s : real(0);
  const real idx = s / mp.scaled_dt;
  int i = (int)floor(idx);
  i = i < 0 ? 0 : (i > mp.M - 2 ? mp.M - 2 : i);
  i0 = i;
  w = idx - (real)i;
}

template <typename real>
__device__ inline real mp_lerp(real a, real b, real w) {
  const real d = b - a;
  return (fabs(w) < real(0.5)) ? a + w * d : b - d * (real(1) - w);
}

template <typename real>
__device__ inline void prodmp_row(const MPParams<real>& mp, real t, real t0, real* row) {
  const int C = 4 + 2 * mp.nbg;
  int i0, j0;
  real w, wi;
  mp_lerp_setup(mp, t, i0, w);
  mp_lerp_setup(mp, t0, j0, wi);
  const real* ra = mp.tab + (int64_t)i0 * C;
  const real* rb = ra + C;
  const real* qa = mp.tab + (int64_t)j0 * C;
  const real* qb = qa + C;
  const real y1 = mp_lerp(ra[0], rb[0], w), y2 = mp_lerp(ra[1], rb[1], w);
  const real dy1 = mp_lerp(ra[2], rb[2], w), dy2 = mp_lerp(ra[3], rb[3], w);
  const real y1i = mp_lerp(qa[0], qb[0], wi), y2i = mp_lerp(qa[1], qb[1], wi);
  const real dy1i = mp_lerp(qa[2], qb[2], wi), dy2i = mp_lerp(qa[3], qb[3], wi);
  const real det = y1i * dy2i - y2i * dy1i;
  const real xi1 = (dy2i / det) * y1 - (dy1i / det) * y2;
  const real xi2 = (y1i / det) * y2 - (y2i / det) * y1;
  const real xi3 = (dy2i / det) * dy1 - (dy1i / det) * dy2;
  const real xi4 = (y1i / det) * dy2 - (y2i / det) * dy1;
  real hgp = 0, hgv = 0;
#pragma unroll
  for (int b = 0; b < TCE_MAXB; ++b) {
    real hp = 0, hv = 0;
    if (b < mp.nbg) {
      const real P = mp_lerp(ra[4 + b], rb[4 + b], w);
      const real V = mp_lerp(ra[4 + mp.nbg + b], rb[4 + mp.nbg + b], w);
      const real Pi = mp_lerp(qa[4 + b], qb[4 + b], wi);
      const real Vi = mp_lerp(qa[4 + mp.nbg + b], qb[4 + mp.nbg + b], wi);
      hp = P - xi1 * Pi - xi2 * Vi;
      hv = V - xi3 * Pi - xi4 * Vi;
      if (b == mp.nbg - 1) { hgp = hp * mp.inv_scale_g; hgv = hv * mp.inv_scale_g; }
    }
    row[4 + b] = hp;
    row[4 + TCE_MAXB + b] = hv / mp.tau;
  }
  const real rel = mp.rel_goal ? real(1) : real(0);
  row[0] = xi1 + rel * hgp;
  row[1] = xi2 * mp.tau;
  row[2] = (xi3 + rel * hgv) / mp.tau;
  row[3] = xi4;
}

// Compact global layout of a basis row: [c0..c3, Hpos[nbg], Hvel[nbg]].
template <typename real>
__device__ inline void mp_row_store(const real* row, int nbg, real* dst) {
#pragma unroll
  for (int i = 0; i < 4; ++i) dst[i] = row[i];
#pragma unroll
  for (int b = 0; b < TCE_MAXB; ++b)
    if (b < nbg) { dst[4 + b] = row[4 + b]; dst[4 + nbg + b] = row[4 + TCE_MAXB + b]; }
}
template <typename real>
__device__ inline void mp_row_load(const real* src, int nbg, real* row) {
#pragma unroll
  for (int i = 0; i < 4; ++i) row[i] = src[i];
#pragma unroll
  for (int b = 0; b < TCE_MAXB; ++b) {
    row[4 + b] = b < nbg ? src[4 + b] : real(0);
    row[4 + TCE_MAXB + b] = b < nbg ? src[4 + nbg + b] : real(0);
  }
}

// torch.linspace element i of `steps` between a and b, in `real` arithmetic
// (aten RangeFactories: symmetric two-sided formula), un-fused like torch.
template <typename real>
__device__ inline real torch_linspace_at(real a, real b, int steps, int i) {
#pragma clang fp contract(off)
  if (steps == 1) return a;
  const real step = (b - a) / (real)(steps - 1);
  const int half = steps / 2;
  return i < half ? a + step * (real)i : b - step * (real)(steps - 1 - i);
}

// times[n, i] of TemporalCorrelatedSampler.get_times
// (mprl/rl/sampler/temporal_correlated_sampler.py:64-78 ->
//  mprl/util/util_matrix.py:139-192): w_s[i]*(t0+dt) + w_e[i]*(t0+T*dt).
// off_first = dt and off_last = T*dt are rounded on the host exactly as the
// reference's Python scalars are.
template <typename real>
__device__ inline real sampler_time_at(real t0, real off_first, real off_last, int T, int i) {
#pragma clang fp contract(off)
  // the reference builds the weights with a default-dtype (float32)
  // torch.linspace and only then casts them `.to(start)` (util_matrix.py:181-184)
  const real ws = (real)torch_linspace_at<float>(1.0f, 0.0f, T, i);
  const real we = (real)torch_linspace_at<float>(0.0f, 1.0f, T, i);
  const real start = t0 + off_first;
  const real end = t0 + off_last;
  const real p1 = ws * start;
  const real p2 = we * end;
  return p1 + p2;
}

// Basis rows for the time grid of env 0 + "init times are not all equal" flag.
// times_row: explicit [T] row (row 0 of the caller's times tensor).  The flag
// is written (not accumulated) by workgroup 0, so no memset is needed.
template <typename real>
__global__ __launch_bounds__(256) void prodmp_basis_kernel(
    MPParams<real> mp, const real* __restrict__ times_row,
    const real* __restrict__ init_time, int64_t N, int T,
    real* __restrict__ B, int* __restrict__ nonuniform) {
  const real t0 = init_time[0];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < T) {
    real row[TCE_ROWLEN];
    prodmp_row(mp, times_row[i], t0, row);
    mp_row_store(row, mp.nbg, B + (int64_t)i * (4 + 2 * mp.nbg));
  }
  if (blockIdx.x == 0) {
    __shared__ int s_bad;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    bool bad = false;
    for (int64_t n = threadIdx.x; n < N; n += 256) bad |= (init_time[n] != t0);
    if (bad) s_bad = 1;
    __syncthreads();
    if (threadIdx.x == 0) *nonuniform = s_bad;
  }
}
