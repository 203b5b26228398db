// Pair-wise trajectory log-probability (forward + backward) for gfx950.
//
// Replaces TemporalCorrelatedPolicy.log_prob
//   (mprl/rl/policy/temporal_correlated_policy.py:104-203): for each (env n,
// pair p=(a,b)) the Gaussian over the R = 2*dof positions at the two pair
// times, dof-major (d0@a, d0@b, d1@a, ...):
//     y  = traj[n, {a,b}, :dof]
//     mu = H_p theta_n + c_p(y0, v0),  C = (H_p L_n)(H_p L_n)^T + reg I
//     logp = -1/2 |Lc^-1 (y-mu)|^2 - sum log diag Lc - R/2 log 2pi
// The reference materialises H Sigma H^T [N,P,R,R] and the expanded
// [N,P,K,K] Cholesky factors; here one workgroup owns one env, keeps L_n in
// LDS and walks the pairs in chunks: M = H_p L (block structure of H_p: row
// (d,j) only touches rows d*nbg..d*nbg+nbg-1 of L), C = M M^T, an R x R
// Cholesky per pair, all in LDS.  The backward kernel recomputes the forward
// (nothing but logp is stored) and produces dL/dmean [N,K] and dL/dL [N,K,K]:
//     alpha = C^-1 (y-mu),  dmu = g alpha,  G = g (alpha alpha^T - C^-1)
//     dmean += H_p^T dmu,   dL += H_p^T (G M)       (lower triangle)
// Small-matrix VALU/LDS work, not HBM-bound: L_n (K^2) is read once per env.
#include "prodmp.h"
#include "mfma16.h"

namespace {

constexpr int PL_BT = 256;
constexpr int PL_MAXR = 16;   // 2 * dof <= 16

// odd row pitch: K + 1 is even for odd K (K = 63 -> 64: a whole column in one bank)
__host__ __device__ inline int pl_pitch(int K) { return (K + 1) | 1; }

struct PLShape {
  int K, R, P, PC, nbg, dof;
};

// LDS carve (in reals): Ls[K][K+1] | ms[K] | Hs[P][2][nbg] | cs[P][2][2] |
// Ms[PC][R][K+1] | Cs[PC][R][R+1] | dv[PC][R] | Li[PC][R][R+1] | al[PC][R] | gs[PC]
__host__ __device__ inline size_t pl_lds_reals(const PLShape& s, bool bwd) {
  size_t n = (size_t)s.K * pl_pitch(s.K) + s.K + (size_t)s.P * 2 * s.nbg + (size_t)s.P * 4;
  n += (size_t)s.PC * s.R * pl_pitch(s.K) + (size_t)s.PC * s.R * (s.R + 1) + (size_t)s.PC * s.R;
  if (bwd) n += (size_t)s.PC * s.R * (s.R + 1) + (size_t)s.PC * s.R + s.PC;
  return n;
}

// io = the dtype of the buffers, real = double: as in pair_prep_kernel below, the
// small-matrix arithmetic (M, C, Cholesky, solves, and with them the backward
// products) runs in double also for float32 buffers -- C is nearly singular and
// float32 there costs 1 - 3e-5 of max |logp|.  The basis rows
// stay in the buffers' precision (they are not what limits it).
template <typename io, typename real, bool BWD>
__global__ __launch_bounds__(PL_BT) void pair_logprob_kernel(
    const io* __restrict__ traj, const io* __restrict__ mean,
    const io* __restrict__ L, int64_t sL, const int64_t* __restrict__ pairs,
    const io* __restrict__ B, const int* __restrict__ nonuniform,
    MPParams<io> mp, const io* __restrict__ times, int times_general,
    const io* __restrict__ t0, const io* __restrict__ y0,
    const io* __restrict__ v0, io reg_io, io* __restrict__ logp,
    const io* __restrict__ gout, io* __restrict__ gmean,
    io* __restrict__ gL, int T, PLShape s, int skip_if_uniform) {
  const real reg = (real)reg_io;
  if (skip_if_uniform && !times_general && *nonuniform == 0) return;   // fast path ran
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  real* smem = reinterpret_cast<real*>(smem_raw);
  const int K = s.K, R = s.R, P = s.P, PC = s.PC, nbg = s.nbg, dof = s.dof;
  const int KP = pl_pitch(K), RP = R + 1;
  real* Ls = smem;
  real* ms = Ls + K * KP;
  real* Hs = ms + K;
  real* cs = Hs + P * 2 * nbg;
  real* Ms = cs + P * 4;
  real* Cs = Ms + PC * R * KP;
  real* dv = Cs + PC * R * RP;
  real* Li = dv + PC * R;                 // BWD only
  real* al = Li + (BWD ? PC * R * RP : 0);
  real* gs = al + (BWD ? PC * R : 0);

  const int tid = threadIdx.x;
  const int64_t n = blockIdx.x;
  const io* Ln = L + n * sL;
  const bool general = times_general || (*nonuniform != 0);

  for (int i = tid; i < K * K; i += PL_BT) {
    const int r = i / K, c = i - r * K;
    Ls[r * KP + c] = (c <= r) ? (real)Ln[i] : real(0);
  }
  for (int i = tid; i < K; i += PL_BT) ms[i] = (real)mean[n * K + i];
  for (int i = tid; i < P * 2; i += PL_BT) {
    const int ti = (int)pairs[i];          // pairs[p][j], i = 2p + j
    io row[TCE_ROWLEN];
    if (general) prodmp_row(mp, times[n * T + ti], t0[n], row);
    else mp_row_load(B + (int64_t)ti * (4 + 2 * nbg), nbg, row);
#pragma unroll
    for (int b = 0; b < TCE_MAXB; ++b)
      if (b < nbg) Hs[i * nbg + b] = (real)row[4 + b];
    cs[i * 2 + 0] = (real)row[0];
    cs[i * 2 + 1] = (real)row[1];
  }

  // backward accumulators: entry e = tid + i*256 of the K x K gradient
  real accL[16];
  real accm = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) accL[i] = 0;
  __syncthreads();

  for (int p0 = 0; p0 < P; p0 += PC) {
    const int npc = min(PC, P - p0);
    // 1. M = H_p L (lower triangle of L), d = y - mu
    for (int i = tid; i < npc * R * K; i += PL_BT) {
      const int k = i % K;
      const int pr = i / K;
      const int r = pr % R, pc = pr / R;
      const int d = r >> 1, j = r & 1;
      const real* h = Hs + ((p0 + pc) * 2 + j) * nbg;
      real acc = 0;
      for (int b = 0; b < nbg; ++b) acc += h[b] * Ls[(d * nbg + b) * KP + k];
      Ms[(pc * R + r) * KP + k] = acc;
    }
    for (int i = tid; i < npc * R; i += PL_BT) {
      const int r = i % R, pc = i / R;
      const int d = r >> 1, j = r & 1;
      const int pj = (p0 + pc) * 2 + j;
      const real* h = Hs + pj * nbg;
      real mu = cs[pj * 2] * (real)y0[n * dof + d] + cs[pj * 2 + 1] * (real)v0[n * dof + d];
      for (int b = 0; b < nbg; ++b) mu += h[b] * ms[d * nbg + b];
      const real y = (real)traj[(n * T + (int)pairs[pj]) * (int64_t)(2 * dof) + d];
      dv[i] = y - mu;
    }
    if (BWD) for (int i = tid; i < npc; i += PL_BT) gs[i] = (real)gout[n * P + p0 + i];
    __syncthreads();
    // 2. C = M M^T + reg I (lower triangle)
    for (int i = tid; i < npc * R * R; i += PL_BT) {
      const int c = i % R;
      const int pr = i / R;
      const int r = pr % R, pc = pr / R;
      if (c <= r) {
        const real* a = Ms + (pc * R + r) * KP;
        const real* b = Ms + (pc * R + c) * KP;
        real acc = 0;
        for (int k = 0; k < K; ++k) acc += a[k] * b[k];
        if (c == r) acc += reg;
        Cs[(pc * R + r) * RP + c] = acc;
      }
    }
    __syncthreads();
    // 3. per pair: Cholesky (in place), z = Lc^-1 d, logp
    if (tid < npc) {
      real* C = Cs + tid * R * RP;
      real* d = dv + tid * R;
      real logdet = 0, quad = 0;
      for (int jx = 0; jx < R; ++jx) {
        real sdiag = C[jx * RP + jx];
        for (int k = 0; k < jx; ++k) sdiag -= C[jx * RP + k] * C[jx * RP + k];
        const real ljj = sqrt(sdiag);
        C[jx * RP + jx] = ljj;
        const real inv = real(1) / ljj;
        for (int i = jx + 1; i < R; ++i) {
          real v = C[i * RP + jx];
          for (int k = 0; k < jx; ++k) v -= C[i * RP + k] * C[jx * RP + k];
          C[i * RP + jx] = v * inv;
        }
        real z = d[jx];
        for (int k = 0; k < jx; ++k) z -= C[jx * RP + k] * d[k];
        z *= inv;
        d[jx] = z;                       // d now holds z
        quad += z * z;
        logdet += log(ljj);
      }
      if (!BWD)
        logp[n * P + p0 + tid] =
            (io)(real(-0.5) * quad - logdet - real(0.5) * (real)R * real(1.8378770664093453));
    }
    if (BWD) {
      __syncthreads();
      // 3b. Linv column c of pair pc (forward substitution on e_c), alpha = Lc^-T z
      for (int i = tid; i < npc * R; i += PL_BT) {
        const int c = i % R, pc = i / R;
        const real* C = Cs + pc * R * RP;
        real* X = Li + pc * R * RP;
        for (int r = 0; r < R; ++r) {
          real v = (r == c) ? real(1) : real(0);
          for (int k = c; k < r; ++k) v -= C[r * RP + k] * X[k * RP + c];
          X[r * RP + c] = (r < c) ? real(0) : v / C[r * RP + r];
        }
      }
      __syncthreads();
      for (int i = tid; i < npc * R; i += PL_BT) {
        const int r = i % R, pc = i / R;
        const real* X = Li + pc * R * RP;
        const real* z = dv + pc * R;
        real a = 0;
        for (int m = r; m < R; ++m) a += X[m * RP + r] * z[m];   // (Linv^T z)_r
        al[i] = a;
      }
      __syncthreads();
      // G = g (alpha alpha^T - Linv^T Linv) -> Cs (full R x R)
      for (int i = tid; i < npc * R * R; i += PL_BT) {
        const int c = i % R;
        const int pr = i / R;
        const int r = pr % R, pc = pr / R;
        const real* X = Li + pc * R * RP;
        real cinv = 0;
        for (int m = max(r, c); m < R; ++m) cinv += X[m * RP + r] * X[m * RP + c];
        Cs[(pc * R + r) * RP + c] = gs[pc] * (al[pc * R + r] * al[pc * R + c] - cinv);
      }
      __syncthreads();
      // 4. GM = G M, in place per column k
      for (int i = tid; i < npc * K; i += PL_BT) {
        const int k = i % K, pc = i / K;
        real m[PL_MAXR];
#pragma unroll
        for (int r = 0; r < PL_MAXR; ++r) m[r] = r < R ? Ms[(pc * R + r) * KP + k] : real(0);
#pragma unroll
        for (int r = 0; r < PL_MAXR; ++r) {
          if (r < R) {
            const real* g = Cs + (pc * R + r) * RP;
            real acc = 0;
#pragma unroll
            for (int c = 0; c < PL_MAXR; ++c)
              if (c < R) acc += g[c] * m[c];
            Ms[(pc * R + r) * KP + k] = acc;
          }
        }
      }
      __syncthreads();
      // 5. dL[row][k] += sum_pc sum_j H[p][j][b] GM[pc][(d,j)][k]; dmean likewise
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int e = tid + i * PL_BT;
        if (e < K * K) {
          const int row = e / K, k = e - row * K;
          if (k <= row) {
            const int d = row / nbg, b = row - d * nbg;
            real acc = 0;
            for (int pc = 0; pc < npc; ++pc) {
              const real* h = Hs + (p0 + pc) * 2 * nbg;
              acc += h[b] * Ms[(pc * R + 2 * d) * KP + k] +
                     h[nbg + b] * Ms[(pc * R + 2 * d + 1) * KP + k];
            }
            accL[i] += acc;
          }
        }
      }
      if (tid < K) {
        const int d = tid / nbg, b = tid - d * nbg;
        real acc = 0;
        for (int pc = 0; pc < npc; ++pc) {
          const real* h = Hs + (p0 + pc) * 2 * nbg;
          acc += gs[pc] * (h[b] * al[pc * R + 2 * d] + h[nbg + b] * al[pc * R + 2 * d + 1]);
        }
        accm += acc;
      }
    }
    __syncthreads();
  }
  if (BWD) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = tid + i * PL_BT;
      if (e < K * K) gL[n * (int64_t)K * K + e] = (io)accL[i];
    }
    if (tid < K) gmean[n * K + tid] = (io)accm;
  }
}

// ---------------------------------------------------------------------------
// Shared-L fast path (non-contextual covariance + common init time, i.e. every
// shipped TCE config): M = H_p L, C, its Cholesky factor and inverse depend on
// the pair only, not on the env.
//   pair_prep   (1 block / pair) : M_p, Linv_p, Cinv_p, logdet_p -> workspace
//   pair_env    (lane = env, wave = pair subset): d, z = Linv d, logp; bwd: alpha = Linv^T z,
//                                  dmean, per-block partials of
//                                  S_p = sum_n g alpha alpha^T and sum_n g
//   pair_final  (1 block)        : dL = sum_p H_p^T (S_p - sg_p Cinv_p) M_p
// All three exit immediately when the device flag says the init times differ
// (then the general kernel above does the work).
// ---------------------------------------------------------------------------
inline int64_t tce_sum_dim0_slices_impl(int64_t N, int64_t M) {
  const int64_t col_blocks = ceil_div(M, 64);
  int64_t s = ceil_div(2048, col_blocks);
  s = tmin<int64_t>(s, ceil_div(N, 16));
  return tmax<int64_t>(s, 1);
}

struct PFShape { int K, R, P, nbg, dof; };
int g_pair_env_static = 1;  // tce_pair_env_static(0): the general kernel for every shape (A / B, tests)
int g_cu_budget = 0;       // tce_set_cu_budget: compute units the caller expects to be free (0: all)
inline int pl_cu_count() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess)
      n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}
__host__ __device__ inline int pf_ws_pair(const PFShape& s) {          // reals per pair
  return 2 * s.nbg + 4 + s.R * s.K + 2 * s.R * s.R + 1;
}

// (L and M = H_p L are staged in LDS: read from global memory inside the product
// loops -- as until round 3 -- every term was an L2 round trip of its own, K of
// them in a row per element of C; the kernel sits on the critical path of every
// policy epoch: 67 us at K 24, 91 us at K 63 beside the critic)
// Everything between the factor L and the pair's record -- M = H_p L,
// C = M M^T + reg I, its Cholesky factor, the inverse factor, C^-1, log det --
// is formed in DOUBLE, also for float32 I/O (round 6): C is nearly singular by
// construction (reg = 1e-4 under variances of order one, and the trajectory
// variance vanishes towards the initial condition), and forming / factoring it
// in float32 is what put the float32 log-prob 1 - 3e-5 of max |logp| away from
// the float64 value (scripts/probe_logp_tol.py; the reference's own float32
// result sits there too).  With this kernel in double the record's rounding to
// float32 and the float32 per-env arithmetic leave <= 6e-7.
// 24 workgroups per launch: the double arithmetic costs nothing measurable.
template <typename real>
__global__ __launch_bounds__(256) void pair_prep_kernel(
    const real* __restrict__ L, const int64_t* __restrict__ pairs, const real* __restrict__ B,
    const int* __restrict__ nonuniform, real reg, real* __restrict__ ws, PFShape s) {
  if (*nonuniform != 0) return;
  typedef double wide;
  extern __shared__ __attribute__((aligned(16))) char prep_raw[];
  __shared__ wide C[PL_MAXR][PL_MAXR + 1];
  __shared__ wide X[PL_MAXR][PL_MAXR + 1];
  __shared__ wide Hl[2 * TCE_MAXB];
  const int K = s.K, R = s.R, nbg = s.nbg;
  const int KP = pl_pitch(K);
  wide* Ls = reinterpret_cast<wide*>(prep_raw);     // [K][KP]
  wide* Ms = Ls + K * KP;                            // [R][KP]
  const int p = blockIdx.x, tid = threadIdx.x;
  real* w = ws + (int64_t)p * pf_ws_pair(s);
  real* Hs = w;                       // [2][nbg]
  real* cs = Hs + 2 * nbg;            // [2][2]
  real* M = cs + 4;                   // [R][K]
  real* Li = M + R * K;               // [R][R]
  real* Ci = Li + R * R;              // [R][R]
  real* ld = Ci + R * R;              // logdet
  for (int e = tid; e < K * K; e += 256) {
    const int r = e / K, c = e - r * K;
    Ls[r * KP + c] = (wide)L[e];
  }
  if (tid < 2 * nbg) {
    const int j = tid / nbg, b = tid - j * nbg;
    const real h = B[pairs[2 * p + j] * (4 + 2 * nbg) + 4 + b];
    Hs[tid] = h;
    Hl[tid] = (wide)h;
  }
  if (tid < 4) cs[tid] = B[pairs[2 * p + (tid >> 1)] * (4 + 2 * nbg) + (tid & 1)];
  __syncthreads();
  for (int e = tid; e < R * K; e += 256) {
    const int r = e / K, k = e - r * K;
    const int d = r >> 1, j = r & 1;
    wide acc = 0;
    for (int b = 0; b < nbg; ++b) {
      const int row = d * nbg + b;
      if (k <= row) acc += Hl[j * nbg + b] * Ls[row * KP + k];
    }
    Ms[r * KP + k] = acc;
    M[e] = (real)acc;
  }
  __syncthreads();
  for (int e = tid; e < R * R; e += 256) {
    const int r = e / R, c = e - r * R;
    wide acc = 0;
    for (int k = 0; k < K; ++k) acc += Ms[r * KP + k] * Ms[c * KP + k];
    C[r][c] = acc + (r == c ? (wide)reg : wide(0));
  }
  __syncthreads();
  if (tid < 64) {
    // Cholesky, column by column, lane i = row i (R <= 16): the same sums in the
    // same order as one thread would form them, R dependent steps instead of
    // R^3 / 6 (the kernel sits on the critical path of every policy epoch)
    wide logdet = 0;
    for (int j = 0; j < R; ++j) {
      wide v = 0;
      if (tid >= j && tid < R) {
        v = C[tid][j];
        for (int k = 0; k < j; ++k) v -= C[tid][k] * C[j][k];
      }
      const wide ljj = sqrt(__shfl(v, j, 64));
      logdet += log(ljj);
      if (tid == j) C[j][j] = ljj;
      else if (tid > j && tid < R) C[tid][j] = v / ljj;
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
    if (tid == 0) ld[0] = (real)logdet;
  }
  __syncthreads();
  if (tid < R) {                       // column tid of Lc^-1
    const int c = tid;
    for (int r = 0; r < R; ++r) {
      wide v = (r == c) ? wide(1) : wide(0);
      for (int k = c; k < r; ++k) v -= C[r][k] * X[k][c];
      X[r][c] = (r < c) ? wide(0) : v / C[r][r];
    }
  }
  __syncthreads();
  for (int e = tid; e < R * R; e += 256) {
    const int r = e / R, c = e - r * R;
    Li[e] = (real)X[r][c];
    wide acc = 0;
    for (int m = (r > c ? r : c); m < R; ++m) acc += X[m][r] * X[m][c];
    Ci[e] = (real)acc;
  }
}

// Block = EB (<= 64) envs x NW waves (blockDim / 64, <= 16); wave q takes the
// pairs q, q + NW, ...; lane = env.  What is the same for every env of a pair
// -- the two basis rows, the init-condition coefficients, the inverse Cholesky
// factor of the pair covariance (pair_prep_kernel's record) -- is read with
// SCALAR loads straight from that record (the pair index is made wave-uniform):
// the FMAs take it as their scalar operand and no LDS instruction is issued
// for it (round 3 kept the record in LDS: 460 of the ~1600 LDS operations per
// env and pair were broadcast reads of it).  The per-pair reduction
// S_p = sum_env g alpha alpha^T over the block's envs is a [R x EB] x [EB x R]
// product on the exact 16x16x4 matrix instruction: alpha goes through a
// wave-private LDS slab (pitch 17, columns R..15 zero), EB / 4 steps of
// {2 LDS reads, 1 MFMA} per pair (round 3: 3 EB LDS reads and 2 EB FMAs per
// lane, four times per pair -- half of the backward kernel).  The pair-sum of
// dmean over the waves goes through LDS at the end.
constexpr int PE_RP = 17;               // pitch of the alpha slab
template <typename real, bool BWD>
__global__ __launch_bounds__(1024) void pair_env_kernel(
    const real* __restrict__ traj, const real* __restrict__ mean,
    const int64_t* __restrict__ pairs, const int* __restrict__ nonuniform,
    const real* __restrict__ y0, const real* __restrict__ v0, const real* __restrict__ ws,
    real* __restrict__ logp, const real* __restrict__ gout, real* __restrict__ gmean,
    real* __restrict__ spart /* [gridDim.x][P][R*R + 1] */, int64_t N, int T, PFShape s,
    int EB, const real* __restrict__ lp_old, const real* __restrict__ adv, real inv_m) {
  if (*nonuniform != 0) return;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  real* smem = reinterpret_cast<real*>(smem_raw);
  const int K = s.K, R = s.R, P = s.P, nbg = s.nbg, dof = s.dof;
  const int KP = pl_pitch(K);
  const int wsp = pf_ws_pair(s);
  real* ms = smem;                               // [EB][KP]      mean_n
  const int NW = blockDim.x >> 6, NT = blockDim.x;
  real* gmp = ms + EB * KP;                      // [NW][EB][KP]  grad mean per wave (BWD)
  real* Ab = gmp + (BWD ? NW * EB * KP : 0);     // [NW][EB][PE_RP] alpha of the wave's current pair
  real* gs = Ab + (BWD ? NW * EB * PE_RP : 0);   // [NW][EB]
  const int tid = threadIdx.x, q = tid >> 6, e = tid & 63;
  const int64_t n0 = (int64_t)blockIdx.x * EB;
  const int64_t n = n0 + e;
  const bool act = e < EB;
  const bool ok = act && n < N;
  const int64_t nc = n < N ? n : N - 1;
  for (int i = tid; i < EB * K; i += NT) {
    const int en = i / K, k = i - en * K;
    const int64_t nn = n0 + en < N ? n0 + en : N - 1;
    ms[en * KP + k] = mean[nn * K + k];
  }
  if (BWD) {
    for (int i = tid; i < NW * EB * KP; i += NT) gmp[i] = 0;
    for (int i = tid; i < NW * EB * PE_RP; i += NT) Ab[i] = 0;
  }
  __syncthreads();
  real* gmq = gmp + q * EB * KP + e * KP;
  real* Abq = Ab + q * EB * PE_RP;
  real* gsq = gs + q * EB;
  const int qu = __builtin_amdgcn_readfirstlane(q);
  for (int p = qu; p < P; p += NW) {
    const real* __restrict__ Hs = ws + (int64_t)p * wsp;      // scalar loads from here on
    const real* __restrict__ cs = Hs + 2 * nbg;
    const real* __restrict__ Li = cs + 4 + R * K;
    real g = 0;
    if (act) {
      real d[PL_MAXR], z[PL_MAXR];
      const int64_t ta = pairs[2 * p], tb = pairs[2 * p + 1];
#pragma unroll
      for (int r = 0; r < PL_MAXR; ++r) {
        d[r] = 0;
        if (r < R) {
          const int dd = r >> 1, j = r & 1;
          const real* h = Hs + j * nbg;
          real mu = cs[2 * j] * y0[nc * dof + dd] + cs[2 * j + 1] * v0[nc * dof + dd];
          for (int b = 0; b < nbg; ++b) mu += h[b] * ms[e * KP + dd * nbg + b];
          const real y = traj[(nc * T + (j ? tb : ta)) * (int64_t)(2 * dof) + dd];
          d[r] = y - mu;
        }
      }
      real quad = 0;
#pragma unroll
      for (int r = 0; r < PL_MAXR; ++r) {
        z[r] = 0;
        if (r < R) {
          real acc = 0;
#pragma unroll
          for (int c = 0; c < PL_MAXR; ++c)
            if (c <= r) acc += Li[r * R + c] * d[c];
          z[r] = acc;
          quad += acc * acc;
        }
      }
      if (!BWD) {
        if (ok) logp[n * P + p] = real(-0.5) * quad - Li[2 * R * R] -
                                  real(0.5) * (real)R * real(1.8378770664093453);
      } else {
        g = real(0);
        if (ok) {
          if (lp_old) {
            // the surrogate's gradient from the log-prob this kernel has just
            // recomputed (surrogate_kernel's formula): no launch between the
            // forward and the backward pass
            const real lpn = real(-0.5) * quad - Li[2 * R * R] -
                             real(0.5) * (real)R * real(1.8378770664093453);
            const real ra = exp(lpn - lp_old[n * P + p]) * adv[n * P + p];
            g = -ra * inv_m;
            if (logp) logp[n * P + p] = lpn;         // for the loss value (record row)
          } else {
            g = gout[n * P + p];
          }
        }
#pragma unroll
        for (int r = 0; r < PL_MAXR; ++r) {
          if (r < R) {
            real al = 0;
#pragma unroll
            for (int m = 0; m < PL_MAXR; ++m)
              if (m >= r && m < R) al += Li[m * R + r] * z[m];
            Abq[e * PE_RP + r] = al;
            const int dd = r >> 1, j = r & 1;
            const real* h = Hs + j * nbg;
            const real ga = g * al;
            for (int b = 0; b < nbg; ++b) gmq[dd * nbg + b] += h[b] * ga;
          }
        }
        gsq[e] = g;
      }
    }
    if (BWD) {
      // wave-private slab: LDS operations of one wave complete in order
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // S[r][c] = sum_env g alpha_r alpha_c: A[m = r][k = env] = g alpha, B[k = env][n = c] = alpha
      typename Mfma16<real>::acc S = {0, 0, 0, 0};
      const int x = e & 15, kq = e >> 4;
      for (int ks = 0; ks < EB / 4; ++ks) {
        const int en = 4 * ks + kq;
        const real b = Abq[en * PE_RP + x];
        S = mfma16(b * gsq[en], b, S);
      }
      const real sg = sizeof(real) == 8 ? (real)wave_sum_f64((double)g) : (real)wave_sum(g);
      real* out = spart + ((int64_t)blockIdx.x * P + p) * (R * R + 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = mfma16_row<real>(kq, i);
        if (r < R && x < R) out[r * R + x] = S[i];
      }
      if (e == 0) out[R * R] = sg;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  if (BWD) {
    __syncthreads();
    for (int i = tid; i < EB * K; i += NT) {
      const int en = i / K, k = i - en * K;
      if (n0 + en < N) {
        const real* g0 = gmp + en * KP + k;
        real acc = 0;
        for (int w = 0; w < NW; ++w) acc += g0[w * EB * KP];
        gmean[(n0 + en) * K + k] = acc;
      }
    }
  }
}

// The same kernel for the (dof, num_basis + 1) combinations the shipped configs
// use, with every per-env vector in REGISTERS: the mean parameters of the lane's
// env (K values), and -- backward -- its accumulated mean gradient.  In the
// general kernel above both live in LDS and the gradient is a chain of
// R (nbg) dependent LDS read-modify-writes per pair (the compiler cannot tell
// the runtime-indexed addresses apart): 126 round trips per env and pair at
// dof 7 / 9 basis rows, which is what its 0.57 ms per K 63 epoch were.  Here a
// pair costs ~460 register FMAs, 14 gathered loads and the EB / 4 MFMA steps of
// the S_p reduction.
template <typename real, bool BWD, int DOF, int NBG>
__global__ __launch_bounds__(256) void pair_env_static_kernel(
    const real* __restrict__ traj, const real* __restrict__ mean,
    const int64_t* __restrict__ pairs, const int* __restrict__ nonuniform,
    const real* __restrict__ y0, const real* __restrict__ v0, const real* __restrict__ ws,
    real* __restrict__ logp, const real* __restrict__ gout, real* __restrict__ gmean,
    real* __restrict__ spart /* [gridDim.x][P][R*R + 1] */, int64_t N, int T, int P, int EB,
    const real* __restrict__ lp_old, const real* __restrict__ adv, real inv_m) {
  if (*nonuniform != 0) return;
  constexpr int R = 2 * DOF, K = DOF * NBG, KP = (K + 1) | 1;
  constexpr int wsp = 2 * NBG + 4 + R * K + 2 * R * R + 1;          // pf_ws_pair
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  real* smem = reinterpret_cast<real*>(smem_raw);
  const int NW = blockDim.x >> 6, NT = blockDim.x;
  real* gmp = smem;                              // [NW][EB][KP]: mean rows first, per-wave gradients last (BWD)
  real* Ab = gmp + (BWD ? NW : 1) * EB * KP;     // [NW][EB][PE_RP] alpha of the wave's current pair (BWD)
  real* gs = Ab + (BWD ? NW * EB * PE_RP : 0);   // [NW][EB]
  const int tid = threadIdx.x, q = tid >> 6, e = tid & 63;
  const int64_t n0 = (int64_t)blockIdx.x * EB;
  const int64_t n = n0 + e;
  const bool act = e < EB;
  const bool ok = act && n < N;
  const int64_t nc = n < N ? n : N - 1;
  // the block's mean rows through LDS (coalesced), then each lane's own row
  for (int i = tid; i < EB * K; i += NT) {
    const int en = i / K, k = i - en * K;
    const int64_t nn = n0 + en < N ? n0 + en : N - 1;
    gmp[en * KP + k] = mean[nn * K + k];
  }
  if (BWD)
    for (int i = tid; i < NW * EB * PE_RP; i += NT) Ab[i] = 0;
  __syncthreads();
  real mn[K], gm[BWD ? K : 1];
#pragma unroll
  for (int k = 0; k < K; ++k) mn[k] = act ? gmp[e * KP + k] : real(0);
  if (BWD) {
#pragma unroll
    for (int k = 0; k < K; ++k) gm[k] = 0;
  }
  real yi[DOF], vi[DOF];
#pragma unroll
  for (int d = 0; d < DOF; ++d) { yi[d] = y0[nc * DOF + d]; vi[d] = v0[nc * DOF + d]; }
  __syncthreads();                                 // gmp is reused for the gradients
  real* Abq = Ab + q * EB * PE_RP;
  real* gsq = gs + q * EB;
  const int qu = __builtin_amdgcn_readfirstlane(q);
  for (int p = qu; p < P; p += NW) {
    const real* __restrict__ Hs = ws + (int64_t)p * wsp;      // scalar loads
    const real* __restrict__ cs = Hs + 2 * NBG;
    const real* __restrict__ Li = cs + 4 + R * K;
    const int64_t ta = pairs[2 * p], tb = pairs[2 * p + 1];
    real d[R], z[R];
#pragma unroll
    for (int dd = 0; dd < DOF; ++dd)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        real mu = cs[2 * j] * yi[dd] + cs[2 * j + 1] * vi[dd];
#pragma unroll
        for (int b = 0; b < NBG; ++b) mu += Hs[j * NBG + b] * mn[dd * NBG + b];
        const real y = traj[(nc * T + (j ? tb : ta)) * (int64_t)R + dd];
        d[2 * dd + j] = y - mu;
      }
    real quad = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      real acc = 0;
#pragma unroll
      for (int c = 0; c <= r; ++c) acc += Li[r * R + c] * d[c];
      z[r] = acc;
      quad += acc * acc;
    }
    if (!BWD) {
      if (ok) logp[n * P + p] = real(-0.5) * quad - Li[2 * R * R] -
                                real(0.5) * (real)R * real(1.8378770664093453);
    } else {
      real g = 0;
      if (ok) {
        if (lp_old) {
          const real lpn = real(-0.5) * quad - Li[2 * R * R] -
                           real(0.5) * (real)R * real(1.8378770664093453);
          const real ra = exp(lpn - lp_old[n * P + p]) * adv[n * P + p];
          g = -ra * inv_m;
          if (logp) logp[n * P + p] = lpn;
        } else {
          g = gout[n * P + p];
        }
      }
      real ga[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        real al = 0;
#pragma unroll
        for (int m = r; m < R; ++m) al += Li[m * R + r] * z[m];
        if (act) Abq[e * PE_RP + r] = al;
        ga[r] = g * al;
      }
#pragma unroll
      for (int dd = 0; dd < DOF; ++dd)
#pragma unroll
        for (int b = 0; b < NBG; ++b)
          gm[dd * NBG + b] += Hs[b] * ga[2 * dd] + Hs[NBG + b] * ga[2 * dd + 1];
      if (act) gsq[e] = g;
      // wave-private slab: LDS operations of one wave complete in order
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      typename Mfma16<real>::acc S = {0, 0, 0, 0};
      const int x = e & 15, kq = e >> 4;
      for (int ks = 0; ks < EB / 4; ++ks) {
        const int en = 4 * ks + kq;
        const real b = Abq[en * PE_RP + x];
        S = mfma16(b * gsq[en], b, S);
      }
      const real sg = sizeof(real) == 8 ? (real)wave_sum_f64((double)g) : (real)wave_sum(g);
      real* out = spart + ((int64_t)blockIdx.x * P + p) * (R * R + 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = mfma16_row<real>(kq, i);
        if (r < R && x < R) out[r * R + x] = S[i];
      }
      if (e == 0) out[R * R] = sg;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  if (BWD) {
    if (act) {
#pragma unroll
      for (int k = 0; k < K; ++k) gmp[(q * EB + e) * KP + k] = gm[k];
    }
    __syncthreads();
    for (int i = tid; i < EB * K; i += NT) {
      const int en = i / K, k = i - en * K;
      if (n0 + en < N) {
        const real* g0 = gmp + en * KP + k;
        real acc = 0;
        for (int w = 0; w < NW; ++w) acc += g0[w * EB * KP];
        gmean[(n0 + en) * K + k] = acc;
      }
    }
  }
}

// the (dof, nbg) combinations with a register kernel: Metaworld 5 / 8 basis
// functions (dof 4), box pushing and table tennis 8 (dof 7), table tennis 3
template <typename real, bool BWD>
void* pair_env_static(int dof, int nbg) {
#define PE_CASE(D, B) if (dof == D && nbg == B) return reinterpret_cast<void*>(pair_env_static_kernel<real, BWD, D, B>)
  PE_CASE(4, 6);
  PE_CASE(4, 9);
  PE_CASE(7, 9);
  PE_CASE(7, 4);
#undef PE_CASE
  return nullptr;
}

// One block per pair: contribution of pair p to dL, gLp[p] [K,K] (the caller
// sums over p with sum_dim0).
template <typename real>
__global__ __launch_bounds__(256) void pair_final_kernel(
    const real* __restrict__ spart, int nblk, const int* __restrict__ nonuniform,
    const real* __restrict__ ws, real* __restrict__ gLp, PFShape s) {
  if (*nonuniform != 0) return;
  __shared__ real G[PL_MAXR][PL_MAXR + 1];
  __shared__ real red[4][PL_MAXR * PL_MAXR + 1];
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  real* GM = reinterpret_cast<real*>(smem_raw);      // [R][K]
  const int K = s.K, R = s.R, P = s.P, nbg = s.nbg;
  real* Ms = GM + R * K;                             // [R][K]: M of the pair's record
  const int tid = threadIdx.x, p = blockIdx.x;
  const real* w = ws + (int64_t)p * pf_ws_pair(s);
  const real* Hs = w;
  const real* M = w + 2 * nbg + 4;
  const real* Ci = M + R * K + R * R;
  for (int e = tid; e < R * K; e += 256) Ms[e] = M[e];
  // S_p and sg_p: sum the per-block partials (4 groups of blocks in parallel)
  const int nv = R * R + 1;
  for (int e = tid; e < 4 * nv; e += 256) {
    const int grp = e / nv, i = e - grp * nv;
    // 8 partial sums: 8 loads in flight (one accumulator made this loop a chain
    // of L2 round trips, 36 us at 256 blocks)
    real a8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int b = grp;
    for (; b + 28 < nblk; b += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a8[u] += spart[((int64_t)(b + 4 * u) * P + p) * nv + i];
    }
    for (; b < nblk; b += 4) a8[0] += spart[((int64_t)b * P + p) * nv + i];
    red[grp][i] = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
  }
  __syncthreads();
  if (tid < R * R) {
    const real sv = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    const real sg = red[0][R * R] + red[1][R * R] + red[2][R * R] + red[3][R * R];
    G[tid / R][tid % R] = sv - sg * Ci[tid];
  }
  __syncthreads();
  for (int e = tid; e < R * K; e += 256) {
    const int r = e / K, k = e - r * K;
    real a = 0;
    for (int c = 0; c < R; ++c) a += G[r][c] * Ms[c * K + k];
    GM[e] = a;
  }
  __syncthreads();
  real* out = gLp + (int64_t)p * K * K;
  for (int e = tid; e < K * K; e += 256) {
    const int row = e / K, k = e - row * K;
    real v = 0;
    if (k <= row) {
      const int d = row / nbg, b = row - d * nbg;
      v = Hs[b] * GM[(2 * d) * K + k] + Hs[nbg + b] * GM[(2 * d + 1) * K + k];
    }
    out[e] = v;
  }
}

// out[j] = sum_n x[n, j]   (x [N, M] row-major).  Two stages so that the whole
// chip streams the N*M elements: stage 1 (grid = column tiles x row slices)
// writes partial sums, stage 2 adds the slices.
template <typename real>
__global__ __launch_bounds__(256) void sum_dim0_kernel(const real* __restrict__ x,
                                                       real* __restrict__ out,
                                                       int64_t N, int64_t M, int64_t rows_per,
                                                       const int* __restrict__ run_if_nonzero,
                                                       const int* __restrict__ run_if_zero = nullptr) {
  if (run_if_nonzero && *run_if_nonzero == 0) return;
  if (run_if_zero && *run_if_zero != 0) return;
  __shared__ real part[4][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t col = (int64_t)blockIdx.x * 64 + c;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per;
  const int64_t r1 = tmin<int64_t>(N, r0 + rows_per);
  real acc = 0;
  if (col < M)
    for (int64_t r = r0 + g; r < r1; r += 4) acc += x[r * M + col];
  part[g][c] = acc;
  __syncthreads();
  if (g == 0 && col < M)
    out[(int64_t)blockIdx.y * M + col] = part[0][c] + part[1][c] + part[2][c] + part[3][c];
}

template <typename real>
int pl_launch(bool bwd, const real* traj, const real* mean, const real* L, int64_t sL,
              const int64_t* pairs, const real* tab, int M, int nbg, real tau, real delay,
              real scaled_dt, real inv_scale_g, int rel_goal, const real* times,
              int times_flags, const real* t0, const real* y0, const real* v0, real reg,
              real* logp, const real* gout, real* gmean, real* gL, real* B, int* flag,
              real* work, int64_t N, int T, int P, int dof, hipStream_t stream,
              const real* lp_old = nullptr, const real* adv = nullptr) {
  const int times_general = times_flags & 1;
  const bool basis_ready = (times_flags & 2) != 0;     // B / flag hold this time grid already
  const bool prep_ready = (times_flags & 4) != 0;      // work holds pair_prep of this L already
  const bool uniform_known = (times_flags & 8) != 0;   // the caller checked: all init times equal
  TCE_CHECK_ARG(traj && mean && L && pairs && tab && times && t0 && y0 && v0 && B && flag,
                "pair_logprob: null buffer");
  TCE_CHECK_ARG(bwd ? ((gout || (lp_old && adv)) && gmean && gL) : (logp != nullptr),
                "pair_logprob: null output");
  const real inv_m = real(1) / (real)(N * (int64_t)P);
  TCE_CHECK_ARG(N > 0 && T > 0 && P > 0, "pair_logprob: bad sizes");
  TCE_CHECK_ARG(nbg >= 1 && nbg <= TCE_MAXB && dof >= 1 && 2 * dof <= PL_MAXR,
                "pair_logprob: num_basis + 1 <= 16 and num_dof <= 8");
  const int K = dof * nbg;
  TCE_CHECK_ARG(K <= 64, "pair_logprob: dof * (num_basis + 1) must be <= 64");
  MPParams<real> mp{tab, M, nbg, tau, delay, scaled_dt, inv_scale_g, rel_goal};
  if (!basis_ready) {
    hipLaunchKernelGGL(prodmp_basis_kernel<real>, dim3((unsigned)ceil_div(T, 256)),
                       dim3(256), 0, stream, mp, times, t0, N, T, B, flag);
    TCE_LAUNCH_CHECK();
  }
  // ---- shared-L fast path (kernels self-disable if the init times differ)
  // workspace carve (shared L only): fast-path scratch | per-env dL | sum scratch
  PFShape f{K, 2 * dof, P, nbg, dof};
  // envs per block of the fast path: as many as the LDS budget allows
  auto env_lds_w = [&](int eb, int nw) {
    return ((size_t)eb * pl_pitch(K) * (bwd ? 1 + nw : 1) +
            (bwd ? nw * ((size_t)eb * PE_RP + eb) : 0)) * sizeof(real);
  };
  auto env_lds = [&](int eb) { return env_lds_w(eb, 4); };
  // up to 64 envs (one lane each) per block of 4 waves; fewer when the four
  // per-wave gradient slabs would not fit the LDS (fp64, K = 63)
  int EB = 64;
  while (EB > 8 && env_lds(EB) > 150 * 1024) EB >>= 1;
  // ... and fewer when that leaves compute units without a block (C2: 4096 envs
  // = 64 blocks of 64): the kernel is a chain of dependent steps per wave, so
  // blocks of 16 envs on every unit beat full waves on a quarter of them
  const int cus_free = g_cu_budget > 0 ? g_cu_budget : pl_cu_count();
  while (EB > 16 && ceil_div(N, EB) < cus_free) EB >>= 1;
  // waves per block: 4, or -- few envs per block -- up to 12 so that a wave
  // walks through 2-3 pairs instead of P / 4
  // -- unless the caller has said that most of the chip is busy with something
  // else (tce_set_cu_budget: the critic epochs beside the policy stream); then
  // wave-instructions count, not latency, and 12 quarter-filled waves lose
  // (measured beside the critic: backward 103 -> 147 us, step 111.5 -> 112.6 ms)
  int NWV = 4;
  if (EB <= 16 && (g_cu_budget <= 0 || g_cu_budget >= pl_cu_count() / 2))
    while (NWV < 12 && NWV * 2 < P && env_lds_w(EB, NWV + 4) <= 64 * 1024) NWV += 4;
  const int nblk = (int)ceil_div(N, EB);
  const int64_t fast_len = (int64_t)P * pf_ws_pair(f) + (int64_t)nblk * P * (f.R * f.R + 1) +
                           (bwd ? (int64_t)P * K * K : 0);
  TCE_CHECK_ARG(sL != 0 || work != nullptr, "pair_logprob: workspace missing (shared L)");
  real* gL_env = gL;                                       // per-env L: written in place
  real* sum_ws = nullptr;
  if (sL == 0 && bwd) {
    gL_env = work + fast_len;
    sum_ws = gL_env + N * (int64_t)K * K;
  }
  const bool fast = (sL == 0) && !times_general && N >= 256;
  TCE_CHECK_ARG(!(bwd && lp_old) || (fast && uniform_known),
                "pair_logprob_bwd_sur: needs the shared-factor fast path known to run (shared L, "
                "affine time grid, >= 256 envs, times_general bit 3 set)");
  if (fast) {
    real* wsp = work;                                      // [P][pf_ws_pair]
    real* spart = wsp + (int64_t)P * pf_ws_pair(f);        // [nblk][P][R*R+1]
    if (!prep_ready) {
      hipLaunchKernelGGL(pair_prep_kernel<real>, dim3(P), dim3(256),
                         (size_t)(K + f.R) * pl_pitch(K) * sizeof(double), stream, L, pairs, B,
                         flag, reg, wsp, f);
      TCE_LAUNCH_CHECK();
    }
    const size_t lds = env_lds_w(EB, NWV);
    TCE_CHECK_ARG(lds <= 150 * 1024, "pair_logprob: fast path LDS");
    void* stat = g_pair_env_static ? (bwd ? pair_env_static<real, true>(dof, nbg)
                                           : pair_env_static<real, false>(dof, nbg)) : nullptr;
    if (stat) {
      typedef void (*kern_t)(const real*, const real*, const int64_t*, const int*, const real*,
                             const real*, const real*, real*, const real*, real*, real*, int64_t,
                             int, int, int, const real*, const real*, real);
      kern_t kern = reinterpret_cast<kern_t>(stat);
      // four waves per block (one per SIMD: a lane holds up to 2 K + 3 R values;
      // __launch_bounds__(256)).  LDS: the per-wave gradient rows [4][EB][KP]
      // double as the staging of the mean rows; the forward needs the latter only
      const int NWS = 4;
      const size_t lds_s = ((size_t)(bwd ? NWS : 1) * EB * pl_pitch(K) +
                            (bwd ? NWS * ((size_t)EB * PE_RP + EB) : 0)) * sizeof(real);
      tce_lds_limit(reinterpret_cast<const void*>(kern), lds_s);
      hipLaunchKernelGGL(kern, dim3(nblk), dim3(64 * NWS), lds_s, stream, traj, mean, pairs, flag,
                         y0, v0, (const real*)wsp, logp, gout, gmean, spart, N, T, P, EB, lp_old,
                         adv, inv_m);
      TCE_LAUNCH_CHECK();
    }
    if (bwd) {
      if (!stat) {
      if (lds > 48 * 1024)
        tce_lds_limit(reinterpret_cast<const void*>(pair_env_kernel<real, true>), (size_t)(lds));
      hipLaunchKernelGGL((pair_env_kernel<real, true>), dim3(nblk), dim3(64 * NWV), lds, stream,
                         traj, mean, pairs, flag, y0, v0, wsp, logp, gout, gmean, spart, N, T,
                         f, EB, lp_old, adv, inv_m);
      TCE_LAUNCH_CHECK();
      }
      real* gLp = spart + (int64_t)nblk * P * (f.R * f.R + 1);   // [P][K][K]
      hipLaunchKernelGGL(pair_final_kernel<real>, dim3(P), dim3(256),
                         (size_t)2 * f.R * K * sizeof(real), stream, spart, nblk, flag, wsp, gLp,
                         f);
      TCE_LAUNCH_CHECK();
      // dL = sum_p gLp[p]  (runs only when the fast path did: flag == 0)
      hipLaunchKernelGGL(sum_dim0_kernel<real>, dim3((unsigned)ceil_div((int64_t)K * K, 64)),
                         dim3(256), 0, stream, gLp, gL, (int64_t)P, (int64_t)K * K,
                         (int64_t)P, (const int*)nullptr, flag);
    } else if (!stat) {
      if (lds > 48 * 1024)
        tce_lds_limit(reinterpret_cast<const void*>(pair_env_kernel<real, false>), (size_t)(lds));
      hipLaunchKernelGGL((pair_env_kernel<real, false>), dim3(nblk), dim3(64 * NWV), lds, stream,
                         traj, mean, pairs, flag, y0, v0, wsp, logp, gout, gmean, spart, N, T,
                         f, EB, (const real*)nullptr, (const real*)nullptr, real(0));
    }
    TCE_LAUNCH_CHECK();
  }
  // the general kernel below (and the sum of its per-env dL) exits at once when
  // the fast path ran; a caller that knows it will run saves those launches
  if (fast && uniform_known) return 0;

  PLShape s{K, 2 * dof, P, P, nbg, dof};
  const size_t budget = 60 * 1024;
  while (s.PC > 1 && pl_lds_reals(s, bwd) * sizeof(double) > budget) --s.PC;
  const size_t lds = pl_lds_reals(s, bwd) * sizeof(double);
  TCE_CHECK_ARG(lds <= 150 * 1024, "pair_logprob: problem too large for LDS");
  auto kern = bwd ? pair_logprob_kernel<real, double, true>
                  : pair_logprob_kernel<real, double, false>;
  if (lds > 48 * 1024)
    tce_lds_limit(reinterpret_cast<const void*>(kern), (size_t)(lds));
  TCE_CHECK_ARG(N < (1ll << 31), "pair_logprob: too many envs");
  // with the fast path launched, gL of the general kernel is per env: the fast
  // path writes the already-reduced [K,K] gradient into gL_shared instead
  hipLaunchKernelGGL(kern, dim3((unsigned)N), dim3(PL_BT), lds, stream, traj, mean, L, sL,
                     pairs, B, flag, mp, times, times_general, t0, y0, v0, reg, logp, gout,
                     gmean, gL_env, T, s, fast ? 1 : 0);
  TCE_LAUNCH_CHECK();
  if (sL == 0 && bwd) {
    // dL of the shared matrix = sum over envs of the general kernel's per-env
    // gradients -- only needed when the general kernel actually ran
    const int64_t M2 = (int64_t)K * K;
    const int64_t slices = tce_sum_dim0_slices_impl(N, M2);
    const int64_t rows_per = ceil_div(N, slices);
    const int* cond = fast ? flag : nullptr;
    dim3 grid((unsigned)ceil_div(M2, 64), (unsigned)slices);
    hipLaunchKernelGGL(sum_dim0_kernel<real>, grid, dim3(256), 0, stream, gL_env,
                       slices == 1 ? gL : sum_ws, N, M2, rows_per, cond);
    TCE_LAUNCH_CHECK();
    if (slices > 1) {
      hipLaunchKernelGGL(sum_dim0_kernel<real>, dim3((unsigned)ceil_div(M2, 64)), dim3(256), 0,
                         stream, sum_ws, gL, slices, M2, slices, cond);
      TCE_LAUNCH_CHECK();
    }
  }
  return 0;
}

template <typename real>
int64_t pl_work_len(int64_t N, int P, int dof, int nbg, int64_t sL, bool bwd) {
  if (sL != 0) return 0;
  const int K = dof * nbg;
  PFShape f{K, 2 * dof, P, nbg, dof};
  const int64_t nblk = ceil_div(N, 8);                   // smallest fast-path block
  int64_t n = (int64_t)P * pf_ws_pair(f) + nblk * P * (f.R * f.R + 1);
  if (bwd) n += (int64_t)P * K * K + N * (int64_t)K * K +
                tce_sum_dim0_slices_impl(N, (int64_t)K * K) * K * K;
  return n;
}

}  // namespace

extern "C" {

int64_t tce_sum_dim0_slices(int64_t N, int64_t M) { return tce_sum_dim0_slices_impl(N, M); }

int tce_set_cu_budget(int compute_units) {
  g_cu_budget = compute_units;
  return 0;
}

int tce_pair_env_static(int on) {
  g_pair_env_static = on ? 1 : 0;
  return 0;
}

// (library-internal: the budget for kernels of other translation units)
int tce_cu_budget_value(void) { return g_cu_budget; }

/* workspace (in elements of the dtype) of the pair log-prob calls; 0 for a
 * per-env L */
int64_t tce_pair_logprob_work_len(int64_t N, int P, int dof, int nbg, int64_t L_stride,
                                  int bwd) {
  return pl_work_len<float>(N, P, dof, nbg, L_stride, bwd != 0);
}

#define DEFINE_PL(SFX, REAL)                                                     \
  int tce_pair_logprob_fwd_##SFX(                                                \
      const REAL* traj, const REAL* mean, const REAL* L, int64_t L_stride,       \
      const int64_t* pairs, const REAL* tab, int M, int nbg, REAL tau,           \
      REAL delay, REAL scaled_dt, REAL inv_scale_g, int rel_goal,                \
      const REAL* times, int times_general, const REAL* init_time,               \
      const REAL* init_pos, const REAL* init_vel, REAL reg, REAL* logp,          \
      REAL* basis_ws, int* flag_ws, REAL* work, int64_t N, int T, int P,         \
      int dof, void* stream) {                                                   \
    return pl_launch<REAL>(false, traj, mean, L, L_stride, pairs, tab, M, nbg,   \
                           tau, delay, scaled_dt, inv_scale_g, rel_goal, times,  \
                           times_general, init_time, init_pos, init_vel, reg,    \
                           logp, nullptr, nullptr, nullptr, basis_ws, flag_ws,   \
                           work, N, T, P, dof, (hipStream_t)stream);             \
  }                                                                              \
  int tce_pair_logprob_bwd_##SFX(                                                \
      const REAL* traj, const REAL* mean, const REAL* L, int64_t L_stride,       \
      const int64_t* pairs, const REAL* tab, int M, int nbg, REAL tau,           \
      REAL delay, REAL scaled_dt, REAL inv_scale_g, int rel_goal,                \
      const REAL* times, int times_general, const REAL* init_time,               \
      const REAL* init_pos, const REAL* init_vel, REAL reg,                      \
      const REAL* grad_logp, REAL* grad_mean, REAL* grad_L, REAL* basis_ws,      \
      int* flag_ws, REAL* work, int64_t N, int T, int P, int dof,                \
      void* stream) {                                                            \
    return pl_launch<REAL>(true, traj, mean, L, L_stride, pairs, tab, M, nbg,    \
                           tau, delay, scaled_dt, inv_scale_g, rel_goal, times,  \
                           times_general, init_time, init_pos, init_vel, reg,    \
                           nullptr, grad_logp, grad_mean, grad_L, basis_ws,      \
                           flag_ws, work, N, T, P, dof, (hipStream_t)stream);    \
  }                                                                              \
  /* backward with the surrogate's gradient formed inside: grad_logp[n, p] =      \
     -exp(logp[n, p] - logp_old[n, p]) adv[n, p] / (N P) from the log-prob the    \
     kernel recomputes */                                                         \
  int tce_pair_logprob_bwd_sur_##SFX(                                            \
      const REAL* traj, const REAL* mean, const REAL* L, int64_t L_stride,       \
      const int64_t* pairs, const REAL* tab, int M, int nbg, REAL tau,           \
      REAL delay, REAL scaled_dt, REAL inv_scale_g, int rel_goal,                \
      const REAL* times, int times_general, const REAL* init_time,               \
      const REAL* init_pos, const REAL* init_vel, REAL reg,                      \
      const REAL* logp_old, const REAL* adv, REAL* logp_out, REAL* grad_mean,    \
      REAL* grad_L, REAL* basis_ws, int* flag_ws, REAL* work, int64_t N, int T,  \
      int P, int dof, void* stream) {                                            \
    TCE_CHECK_ARG(logp_old && adv, "pair_logprob_bwd_sur: null buffer");         \
    return pl_launch<REAL>(true, traj, mean, L, L_stride, pairs, tab, M, nbg,    \
                           tau, delay, scaled_dt, inv_scale_g, rel_goal, times,  \
                           times_general, init_time, init_pos, init_vel, reg,    \
                           logp_out, nullptr, grad_mean, grad_L, basis_ws,       \
                           flag_ws, work, N, T, P, dof, (hipStream_t)stream,     \
                           logp_old, adv);                                       \
  }                                                                              \
  /* ws: REAL [tce_sum_dim0_slices(N, M), M] workspace (may be NULL when the  \
     slice count is 1) */                                                        \
  int tce_sum_dim0_##SFX(const REAL* x, REAL* out, REAL* ws, int64_t N,          \
                         int64_t M, void* stream) {                              \
    TCE_CHECK_ARG(x && out && N > 0 && M > 0, "sum_dim0: bad arguments");        \
    const int64_t slices = tce_sum_dim0_slices(N, M);                            \
    TCE_CHECK_ARG(slices == 1 || ws, "sum_dim0: workspace missing");             \
    const int64_t rows_per = ceil_div(N, slices);                                \
    dim3 grid((unsigned)ceil_div(M, 64), (unsigned)slices);                      \
    hipLaunchKernelGGL(sum_dim0_kernel<REAL>, grid, dim3(256), 0,                \
                       (hipStream_t)stream, x, slices == 1 ? out : ws, N, M,     \
                       rows_per, (const int*)nullptr);                           \
    TCE_LAUNCH_CHECK();                                                          \
    if (slices > 1) {                                                            \
      hipLaunchKernelGGL(sum_dim0_kernel<REAL>, dim3((unsigned)ceil_div(M, 64)), \
                         dim3(256), 0, (hipStream_t)stream, ws, out, slices, M,  \
                         slices, (const int*)nullptr);                           \
      TCE_LAUNCH_CHECK();                                                        \
    }                                                                            \
    return 0;                                                                    \
  }

DEFINE_PL(f32, float)
DEFINE_PL(f64, double)

}  // extern "C"
