// ProDMP trajectory generation for gfx950 (MI355X):
//   times grid (k3), parameter sampling w = mean + L eps (k4/k6),
//   uniform-init-time basis table, and the trajectory kernel (k4) that
//   replaces TemporalCorrelatedPolicy.sample
//   (mprl/rl/policy/temporal_correlated_policy.py:34-102 ->
//    mp_pytorch ProDMP.sample_trajectories / get_traj_pos / get_traj_vel).
//
// prodmp_traj_kernel is write-bound: per env it writes T*2*dof reals and reads
// dof*(nb+1) parameters + 2*dof initial conditions.  Lane <-> time step: each
// lane keeps its basis row (4 + 2*(nb+1) values) in registers across the envs
// of its block, the per-env parameters are wave-uniform (scalar loads), and
// the 2*dof outputs of a lane are contiguous in memory (16-byte stores).
#include "prodmp.h"

namespace {

template <typename real>
__global__ __launch_bounds__(256) void times_kernel(
    const real* __restrict__ t0, real off_first, real off_last,
    real* __restrict__ times, int64_t N, int T) {
  const int64_t total = N * (int64_t)T;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / T;
    times[i] = sampler_time_at<real>(t0[n], off_first, off_last, T, (int)(i - n * T));
  }
}

// w[n, i] = mean[n, i] + sum_{j<=i} L[n, i, j] eps[n, j]   (L batch stride sL: 0 = shared)
template <typename real>
__global__ __launch_bounds__(256) void mvn_rsample_kernel(
    const real* __restrict__ mean, const real* __restrict__ L, int64_t sL,
    const real* __restrict__ eps, real* __restrict__ out, int64_t N, int K) {
  const int64_t total = N * (int64_t)K;
  for (int64_t idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t n = idx / K;
    const int i = (int)(idx - n * K);
    const real* Lr = L + n * sL + (int64_t)i * K;
    const real* e = eps + n * K;
    real acc = 0;
    for (int j = 0; j <= i; ++j) acc += Lr[j] * e[j];
    out[idx] = mean[idx] + acc;
  }
}

template <typename real, int NV>
__device__ inline void store_vec(real* dst, const real* src) {
  typedef real vec __attribute__((ext_vector_type(NV), aligned(sizeof(real))));
  vec v;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = src[i];
  *reinterpret_cast<vec*>(dst) = v;
}

// the same with NV * sizeof(real) alignment guaranteed by the caller (one store)
template <typename real, int NV>
__device__ inline void store_vec_aligned(real* dst, const real* src) {
  typedef real vec __attribute__((ext_vector_type(NV), aligned(NV * sizeof(real))));
  vec v;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = src[i];
  *reinterpret_cast<vec*>(dst) = v;
}

// NBG > 0: num_basis + 1 known at compile time (the loops over the basis
// become straight-line FMAs with SGPR parameter operands); NBG == 0: runtime.
// HALF (DOF == 4 only): a lane owns (t, pos | vel) and writes ONE 16-byte
// chunk, so a wave store covers 1 KiB without gaps; otherwise a lane owns a
// whole step and writes its 2*DOF values.
template <typename real, int DOF, int NBG, bool HALF>
__global__ __launch_bounds__(256) void prodmp_traj_kernel(
    const real* __restrict__ B, const int* __restrict__ nonuniform,
    MPParams<real> mp, const real* __restrict__ times, int times_general,
    const real* __restrict__ w, const real* __restrict__ t0,
    const real* __restrict__ y0, const real* __restrict__ v0,
    real* __restrict__ out, int64_t N, int T, int epb, int only_if_general) {
  // only_if_general: the uniform-grid case was done by prodmp_traj_rows_kernel
  if (only_if_general && !times_general && *nonuniform == 0) return;
  const int nbg = NBG > 0 ? NBG : mp.nbg;
  constexpr int MAXB = NBG > 0 ? NBG : TCE_MAXB;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int t = HALF ? (i >> 1) : i;
  const int h = HALF ? (i & 1) : 0;
  const int64_t n_begin = (int64_t)blockIdx.y * epb;
  const int64_t n_end = tmin<int64_t>(N, n_begin + epb);
  const bool general = times_general || (*nonuniform != 0);
  const int tc = t < T ? t : T - 1;
  real row[TCE_ROWLEN];
  if (!general) mp_row_load(B + (int64_t)tc * (4 + 2 * nbg), nbg, row);
  // HALF: this lane's coefficients (c_y0, c_v0, H[b]) of pos (h=0) or vel (h=1)
  real hr[2 + MAXB];
  auto pick_half = [&]() {
    hr[0] = h ? row[2] : row[0];
    hr[1] = h ? row[3] : row[1];
#pragma unroll
    for (int b = 0; b < MAXB; ++b) hr[2 + b] = h ? row[4 + TCE_MAXB + b] : row[4 + b];
  };
  if (HALF) pick_half();
  for (int64_t n = n_begin; n < n_end; ++n) {
    if (general) {
      prodmp_row(mp, times[n * T + tc], t0[n], row);
      if (HALF) pick_half();
    }
    const real* wn = w + n * (int64_t)(DOF * nbg);
    if (HALF) {
      real o[DOF];
#pragma unroll
      for (int d = 0; d < DOF; ++d) {
        real acc = hr[0] * y0[n * DOF + d] + hr[1] * v0[n * DOF + d];
#pragma unroll
        for (int b = 0; b < MAXB; ++b)
          if (b < nbg) acc += hr[2 + b] * wn[d * nbg + b];
        o[d] = acc;
      }
      if (t < T) store_vec<real, DOF>(out + (n * T + t) * (int64_t)(2 * DOF) + DOF * h, o);
    } else {
      real o[2 * DOF];
#pragma unroll
      for (int d = 0; d < DOF; ++d) {
        const real yd = y0[n * DOF + d], vd = v0[n * DOF + d];
        real pos = row[0] * yd + row[1] * vd;
        real vel = row[2] * yd + row[3] * vd;
#pragma unroll
        for (int b = 0; b < MAXB; ++b) {
          if (b < nbg) {
            const real th = wn[d * nbg + b];
            pos += row[4 + b] * th;
            vel += row[4 + TCE_MAXB + b] * th;
          }
        }
        o[d] = pos;
        o[DOF + d] = vel;
      }
      constexpr int C = 2 * DOF;
      if (C % 4 != 0 && C % 2 == 0) {
        // rows of C = 14 (dof 7) values: a lane's row is not a multiple of 16 B,
        // so row-per-lane stores are 8-byte pieces at a C*s-byte stride (25 % of
        // the HBM rate).  The wave's 64 rows are contiguous in memory: stage
        // them in a wave-private LDS slab and store them as consecutive
        // 2-element chunks, lane = chunk (512 B / 1 KiB contiguous per store).
        __shared__ real slab[4][64 * C];
        real* sl = slab[threadIdx.x >> 6];
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int k = 0; k < C; ++k) sl[lane * C + k] = o[k];
        // the slab is private to this wave and LDS operations of one wave
        // execute in order: only the COMPILER must not reorder across here.  (A
        // release fence also waits for the wave's outstanding global stores --
        // one HBM round trip per env.)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int t_w = t - lane;                              // first step of the wave
        real* wbase = out + (n * T + t_w) * (int64_t)C;
        const int nvalid = (T - t_w < 64 ? T - t_w : 64) * C;  // elements of live rows
#pragma unroll
        for (int q = 0; q < C / 2; ++q) {
          const int e = 2 * (q * 64 + lane);
          if (e < nvalid) store_vec_aligned<real, 2>(wbase + e, sl + e);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      }
      if (t < T) {
        real* dst = out + (n * T + t) * (int64_t)(2 * DOF);
        if (C % 4 == 0) {
#pragma unroll
          for (int k = 0; k < C / 4; ++k) store_vec<real, 4>(dst + 4 * k, o + 4 * k);
        } else if (C % 2 == 0) {
          // handled below (all lanes of the wave take part)
        } else {
#pragma unroll
          for (int k = 0; k < C; ++k) dst[k] = o[k];
        }
      }
    }
  }
}

// Rows of 2 DOF values that are not a multiple of 16 bytes (odd DOF: 7-dof box
// pushing / table tennis, 56-byte rows) on the shared (uniform init time) basis
// table.  Three things differ from prodmp_traj_kernel:
//  * the per-env parameters (w, y0, v0) of a chunk of envs are staged in LDS
//    first, so the env loop issues NO global loads: loads and stores share one
//    in-order counter on gfx950, and a load behind a store waits for the store
//    to reach memory -- one HBM round trip per env (measured: 5.6 us per env and
//    wave, 26 % of the HBM rate);
//  * short horizons (T <= 128, box pushing: 100) put 256 / T consecutive envs
//    side by side in a workgroup (their rows are consecutive in memory), so
//    200 of 256 lanes work instead of 100;
//  * the 64 rows of a wave are contiguous: they go through a wave-private LDS
//    slab and leave as consecutive 2-element chunks, lane = chunk.
template <typename real, int DOF, int NBG>
__global__ __launch_bounds__(256) void prodmp_traj_rows_kernel(
    const real* __restrict__ B, const int* __restrict__ nonuniform, MPParams<real> mp,
    int times_general, const real* __restrict__ w, const real* __restrict__ y0,
    const real* __restrict__ v0, real* __restrict__ out, int64_t N, int T, int spb, int epb) {
  if (times_general || *nonuniform != 0) return;      // prodmp_traj_kernel takes it
  const int nbg = NBG > 0 ? NBG : mp.nbg;
  constexpr int MAXB = NBG > 0 ? NBG : TCE_MAXB;
  constexpr int C = 2 * DOF;
  constexpr int EC = 32;                               // envs per staged chunk
  const int K = DOF * nbg, NPAR = K + 2 * DOF;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  real* par = reinterpret_cast<real*>(smem_raw);       // [EC][NPAR]
  real* slab = par + EC * NPAR;                         // [4 waves][64 * C]
  const int i = threadIdx.x, lane = i & 63;
  int sub = 0, t = blockIdx.x * 256 + i;
  if (spb > 1) { sub = i / T; t = i - sub * T; }
  const bool live = spb > 1 ? sub < spb : t < T;
  const int tc = t < T ? t : T - 1;
  real row[TCE_ROWLEN];
  mp_row_load(B + (int64_t)tc * (4 + 2 * nbg), nbg, row);
  real* sl = slab + (i >> 6) * 64 * C;
  const int64_t n_begin = (int64_t)blockIdx.y * epb;
  const int64_t n_end = tmin<int64_t>(N, n_begin + epb);
  // first global row of this wave relative to the first env of an iteration
  const int r_w = spb > 1 ? (i & ~63) : blockIdx.x * 256 + (i & ~63);
  const int rows_it = spb > 1 ? spb * T : T;            // rows per iteration beyond which nothing lives
  for (int64_t c0 = n_begin; c0 < n_end; c0 += EC) {
    const int ne = (int)tmin<int64_t>(EC, n_end - c0);
    __syncthreads();                                    // the previous chunk is consumed
    for (int idx = i; idx < ne * NPAR; idx += 256) {
      const int e = idx / NPAR, j = idx - e * NPAR;
      const int64_t n = c0 + e;
      par[idx] = j < K ? w[n * K + j] : (j < K + DOF ? y0[n * DOF + j - K] : v0[n * DOF + j - K - DOF]);
    }
    __syncthreads();
    for (int e = 0; e < ne; e += spb) {
      const int slot = e + sub < ne ? e + sub : ne - 1; // dead lanes shadow a live env
      const real* p = par + slot * NPAR;
      real o[C];
#pragma unroll
      for (int d = 0; d < DOF; ++d) {
        const real yd = p[K + d], vd = p[K + DOF + d];
        real pos = row[0] * yd + row[1] * vd;
        real vel = row[2] * yd + row[3] * vd;
#pragma unroll
        for (int b = 0; b < MAXB; ++b) {
          if (b < nbg) {
            const real th = p[d * nbg + b];
            pos += row[4 + b] * th;
            vel += row[4 + TCE_MAXB + b] * th;
          }
        }
        o[d] = pos;
        o[DOF + d] = vel;
      }
#pragma unroll
      for (int k = 0; k < C; ++k) sl[lane * C + k] = o[k];
      asm volatile("" ::: "memory");                    // wave-private slab: compiler order only
      __builtin_amdgcn_wave_barrier();
      const int live_rows = (ne - e < spb ? ne - e : spb) * (spb > 1 ? T : 1);
      const int lim = spb > 1 ? live_rows : (live_rows ? rows_it : 0);
      int nrow = lim - r_w;                             // live rows of this wave
      nrow = nrow < 0 ? 0 : (nrow > 64 ? 64 : nrow);
      real* wbase = out + ((c0 + e) * T + r_w) * (int64_t)C;
      const int nvalid = nrow * C;
      constexpr int CH = C % 4 == 0 ? 4 : 2;            // elements per store (16 B when rows allow)
#pragma unroll
      for (int q = 0; q < C / CH; ++q) {
        const int el = CH * (q * 64 + lane);
        if (el < nvalid) store_vec_aligned<real, CH>(wbase + el, sl + el);
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
  }
  (void)live;
}

template <typename real>
int traj_launch(const real* tab, int M, int nbg, real tau, real delay, real scaled_dt,
                real inv_scale_g, int rel_goal, const real* times, int times_flags,
                const real* w, const real* t0, const real* y0, const real* v0,
                real* out, real* B, int* flag, int64_t N, int T, int dof,
                hipStream_t stream) {
  const int times_general = times_flags & 1;
  const bool basis_ready = (times_flags & 2) != 0;     // B / flag hold this time grid already
  TCE_CHECK_ARG(tab && times && w && t0 && y0 && v0 && out && B && flag,
                "prodmp_traj: null buffer");
  TCE_CHECK_ARG(N > 0 && T > 0 && M >= 2, "prodmp_traj: bad sizes");
  TCE_CHECK_ARG(nbg >= 1 && nbg <= TCE_MAXB, "prodmp_traj: num_basis + 1 must be <= 16");
  TCE_CHECK_ARG(dof >= 1 && dof <= 8, "prodmp_traj: num_dof must be <= 8");
  MPParams<real> mp{tab, M, nbg, tau, delay, scaled_dt, inv_scale_g, rel_goal};
  const int tb = (int)ceil_div(T, 256);
  if (!basis_ready) {
    hipLaunchKernelGGL(prodmp_basis_kernel<real>, dim3((unsigned)ceil_div(T, 256)), dim3(256), 0,
                       stream, mp, times, t0, N, T, B, flag);
    TCE_LAUNCH_CHECK();
  }
  // odd dof (rows not a multiple of 16 B) on the shared basis table: the rows
  // kernel; the kernel below then only runs if the time grid turns out general
  int only_general = 0;
  if (dof == 3 || dof == 4 || dof == 5 || dof == 7) {
    const int spb = T <= 128 ? 256 / T : 1;
    const int txr = spb > 1 ? 1 : (int)ceil_div(T, 256);
    int64_t epr = tmax<int64_t>(spb, (N * txr) / 2048);
    epr = ceil_div(epr, spb) * spb;
    const int64_t gyr = ceil_div(N, epr);
    TCE_CHECK_ARG(gyr <= 65535, "prodmp_traj: too many envs");
    const size_t lds = ((size_t)32 * (dof * nbg + 2 * dof) + (size_t)4 * 64 * 2 * dof) * sizeof(real);
    dim3 gr(txr, (unsigned)gyr);
#define ROWS_GO(D, NB)                                                            \
    hipLaunchKernelGGL((prodmp_traj_rows_kernel<real, D, NB>), gr, dim3(256), lds, \
                       stream, B, flag, mp, times_general, w, y0, v0, out, N, T,   \
                       spb, (int)epr)
    if (dof == 7) {
      if (nbg == 9) ROWS_GO(7, 9);
      else if (nbg == 4) ROWS_GO(7, 4);
      else ROWS_GO(7, 0);
    } else if (dof == 4) {
      if (nbg == 6) ROWS_GO(4, 6);
      else if (nbg == 9) ROWS_GO(4, 9);
      else if (nbg == 5) ROWS_GO(4, 5);
      else ROWS_GO(4, 0);
    } else if (dof == 5) {
      ROWS_GO(5, 0);
    } else {
      ROWS_GO(3, 0);
    }
#undef ROWS_GO
    TCE_LAUNCH_CHECK();
    only_general = 1;
  }
  // ~2048 workgroups: each keeps its basis rows for `epb` envs
  const bool half = (dof == 4);
  const int txb = (int)ceil_div((int64_t)T * (half ? 2 : 1), 256);
  int epb = (int)tmax<int64_t>(1, (N * txb) / 2048);
  const int64_t gy = ceil_div(N, epb);
  TCE_CHECK_ARG(gy <= 65535, "prodmp_traj: too many envs");
  dim3 grid(txb, (unsigned)gy);
#define TRAJ_GO(D, NB, H)                                                         \
  hipLaunchKernelGGL((prodmp_traj_kernel<real, D, NB, H>), grid, dim3(256), 0,    \
                     stream, B, flag, mp, times, times_general, w, t0, y0, v0,    \
                     out, N, T, epb, only_general)
#define TRAJ_NB(D, H)                                                             \
  switch (nbg) {                                                                  \
    case 4: TRAJ_GO(D, 4, H); break;                                              \
    case 5: TRAJ_GO(D, 5, H); break;                                              \
    case 6: TRAJ_GO(D, 6, H); break;                                              \
    case 9: TRAJ_GO(D, 9, H); break;                                              \
    default: TRAJ_GO(D, 0, H); break;                                             \
  }
  switch (dof) {
    case 1: TRAJ_GO(1, 0, false); break;
    case 2: TRAJ_GO(2, 0, false); break;
    case 3: TRAJ_GO(3, 0, false); break;
    case 4: TRAJ_NB(4, true); break;
    case 5: TRAJ_GO(5, 0, false); break;
    case 6: TRAJ_GO(6, 0, false); break;
    case 7: TRAJ_NB(7, false); break;
    default: TRAJ_GO(8, 0, false); break;
  }
#undef TRAJ_NB
#undef TRAJ_GO
  TCE_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" {

#define DEFINE_PRODMP(SFX, REAL)                                                 \
  int tce_times_##SFX(const REAL* t0, REAL off_first, REAL off_last, REAL* times, \
                      int64_t N, int T, void* stream) {                          \
    TCE_CHECK_ARG(t0 && times && N > 0 && T > 0, "times: bad arguments");        \
    const int64_t nb = tmin<int64_t>(ceil_div(N * T, 256), 4096);                \
    hipLaunchKernelGGL(times_kernel<REAL>, dim3((unsigned)nb), dim3(256), 0,     \
                       (hipStream_t)stream, t0, off_first, off_last, times, N, T); \
    TCE_LAUNCH_CHECK();                                                          \
    return 0;                                                                    \
  }                                                                              \
  int tce_mvn_rsample_##SFX(const REAL* mean, const REAL* L, int64_t L_stride,   \
                            const REAL* eps, REAL* out, int64_t N, int K,        \
                            void* stream) {                                      \
    TCE_CHECK_ARG(mean && L && eps && out && N > 0 && K > 0,                     \
                  "mvn_rsample: bad arguments");                                 \
    const int64_t nb = tmin<int64_t>(ceil_div(N * K, 256), 4096);                \
    hipLaunchKernelGGL(mvn_rsample_kernel<REAL>, dim3((unsigned)nb), dim3(256),  \
                       0, (hipStream_t)stream, mean, L, L_stride, eps, out, N, K); \
    TCE_LAUNCH_CHECK();                                                          \
    return 0;                                                                    \
  }                                                                              \
  int tce_prodmp_traj_##SFX(const REAL* tab, int M, int nbg, REAL tau,           \
                            REAL delay, REAL scaled_dt, REAL inv_scale_g,        \
                            int rel_goal, const REAL* times, int times_general,  \
                            const REAL* params, const REAL* init_time,           \
                            const REAL* init_pos, const REAL* init_vel,          \
                            REAL* out, REAL* basis_ws, int* flag_ws, int64_t N,  \
                            int T, int dof, void* stream) {                      \
    return traj_launch<REAL>(tab, M, nbg, tau, delay, scaled_dt, inv_scale_g,    \
                             rel_goal, times, times_general, params, init_time,  \
                             init_pos, init_vel, out, basis_ws, flag_ws, N, T,   \
                             dof, (hipStream_t)stream);                          \
  }

DEFINE_PRODMP(f32, float)
DEFINE_PRODMP(f64, double)

}  // extern "C"
