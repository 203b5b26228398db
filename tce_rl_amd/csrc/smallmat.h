// K x K (K <= 64) dense helpers on LDS-resident double matrices, executed by
// one 256-thread workgroup.  Row pitch KP = sm_pitch(K): the next ODD number
// above K, so that column walks (stride KP doubles) touch all 32 eight-byte
// bank pairs -- K + 1 is even for odd K, and K = 63 (7 dof x 9) made it 64: every
// element of a column in ONE bank (measured: 5 100 cycles per Jacobi round
// instead of 900).  Every helper ends with a barrier unless noted.
#pragma once
#include "common.h"

#define SM_BT 256

__host__ __device__ inline int sm_pitch(int K) { return (K + 1) | 1; }

// C[i][j] = sum_k A[i][k] * B[j][k]        (C = A B^T)
__device__ inline void sm_mm_nt(double* C, const double* A, const double* B, int K, int KP) {
  for (int e = threadIdx.x; e < K * K; e += SM_BT) {
    const int i = e / K, j = e - i * K;
    double acc = 0;
    for (int k = 0; k < K; ++k) acc += A[i * KP + k] * B[j * KP + k];
    C[i * KP + j] = acc;
  }
  __syncthreads();
}
// C[i][j] = sum_k A[i][k] * B[k][j]        (C = A B)
__device__ inline void sm_mm_nn(double* C, const double* A, const double* B, int K, int KP) {
  for (int e = threadIdx.x; e < K * K; e += SM_BT) {
    const int i = e / K, j = e - i * K;
    double acc = 0;
    for (int k = 0; k < K; ++k) acc += A[i * KP + k] * B[k * KP + j];
    C[i * KP + j] = acc;
  }
  __syncthreads();
}
// C[i][j] = sum_k A[k][i] * B[k][j]        (C = A^T B)
__device__ inline void sm_mm_tn(double* C, const double* A, const double* B, int K, int KP) {
  for (int e = threadIdx.x; e < K * K; e += SM_BT) {
    const int i = e / K, j = e - i * K;
    double acc = 0;
    for (int k = 0; k < K; ++k) acc += A[k * KP + i] * B[k * KP + j];
    C[i * KP + j] = acc;
  }
  __syncthreads();
}
// ---- triangular solves: 4 consecutive lanes (a "quad") per column (or row)
// of X.  Row by row: each lane of the quad sums every fourth term of
// sum_k L[r][k] x_k, two quad-permute adds give all four the total, lane 0
// writes x_r back to LDS where the quad reads it in the next rows (the LDS
// executes a wave's operations in order: no barrier inside the solve).  The
// dependent chain per row is one LDS round trip + <= K / 8 FMAs per
// accumulator instead of the K / 2 of a thread-per-column solve.
__device__ inline double sm_quad_sum(double v) {
  v += dpp_perm_f64<0xB1>(v);           // quad_perm [1,0,3,2]
  v += dpp_perm_f64<0x4E>(v);           // quad_perm [2,3,0,1]
  return v;
}
// UPPER = false: X <- L^-1 X (forward);  UPPER = true: X <- L^-T X (backward).
// ROWS = false: the systems are the columns of X;  ROWS = true: its rows
// (UPPER: X <- X L^-1).
template <bool UPPER, bool ROWS>
__device__ inline void sm_tri_solve(double* X, const double* L, int K, int KP) {
  __shared__ double invd[64];
  __syncthreads();                                       // earlier readers of invd are done
  if (threadIdx.x < K) invd[threadIdx.x] = 1.0 / L[threadIdx.x * KP + threadIdx.x];
  __syncthreads();
  const int c = threadIdx.x >> 2, t = threadIdx.x & 3;
  const bool live = c < K;
  const int cc = live ? c : 0;                           // dead quads shadow system 0, never write
  const int xs = ROWS ? 1 : KP;                          // stride between unknowns of a system
  double* xb = X + (ROWS ? cc * KP : cc);
  for (int rr = 0; rr < K; ++rr) {
    const int r = UPPER ? K - 1 - rr : rr;
    double a0 = 0, a1 = 0;
    if (!UPPER) {
      const double* lr = L + r * KP;
      int k = t;
      for (; k + 4 < r; k += 8) {
        a0 += lr[k] * xb[k * xs];
        a1 += lr[k + 4] * xb[(k + 4) * xs];
      }
      if (k < r) a0 += lr[k] * xb[k * xs];
    } else {
      const double* lc = L + r;
      int k = r + 1 + t;
      for (; k + 4 < K; k += 8) {
        a0 += lc[k * KP] * xb[k * xs];
        a1 += lc[(k + 4) * KP] * xb[(k + 4) * xs];
      }
      if (k < K) a0 += lc[k * KP] * xb[k * xs];
    }
    const double acc = sm_quad_sum(a0 + a1);
    const double xr = (xb[r * xs] - acc) * invd[r];
    if (live && t == 0) xb[r * xs] = xr;
    asm volatile("" ::: "memory");                       // program order of the LDS accesses
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
}
// X <- L^-1 X   (L lower triangular)
__device__ inline void sm_trsm_l(double* X, const double* L, int K, int KP) {
  sm_tri_solve<false, false>(X, L, K, KP);
}
// X <- L^-T X
__device__ inline void sm_trsm_lt(double* X, const double* L, int K, int KP) {
  sm_tri_solve<true, false>(X, L, K, KP);
}
// X <- X L^-1   (solve Z L = X: L^T z^T = x^T for every row)
__device__ inline void sm_trsm_r(double* X, const double* L, int K, int KP) {
  sm_tri_solve<true, true>(X, L, K, KP);
}
// In-place Cholesky of the lower triangle of S (right-looking); upper part zeroed.
__device__ inline void sm_cholesky(double* S, int K, int KP) {
  for (int j = 0; j < K; ++j) {
    if (threadIdx.x == 0) S[j * KP + j] = sqrt(S[j * KP + j]);
    __syncthreads();
    const double d = S[j * KP + j];
    for (int i = j + 1 + threadIdx.x; i < K; i += SM_BT) S[i * KP + j] /= d;
    __syncthreads();
    const int n = K - j - 1;
    for (int e = threadIdx.x; e < n * n; e += SM_BT) {
      const int a = e / n, b = e - a * n;
      if (b <= a) S[(j + 1 + a) * KP + j + 1 + b] -= S[(j + 1 + a) * KP + j] * S[(j + 1 + b) * KP + j];
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < K * K; e += SM_BT) {
    const int i = e / K, j = e - i * K;
    if (j > i) S[i * KP + j] = 0;
  }
  __syncthreads();
}

template <typename real>
__device__ inline void sm_load(double* D, const real* __restrict__ src, int K, int KP, bool tril) {
  for (int e = threadIdx.x; e < K * K; e += SM_BT) {
    const int i = e / K, j = e - i * K;
    D[i * KP + j] = (tril && j > i) ? 0.0 : (double)src[e];
  }
  __syncthreads();
}
template <typename real>
__device__ inline void sm_store(real* __restrict__ dst, const double* D, int K, int KP, bool tril,
                                double scale) {
  for (int e = threadIdx.x; e < K * K; e += SM_BT) {
    const int i = e / K, j = e - i * K;
    dst[e] = (tril && j > i) ? real(0) : (real)(scale * D[i * KP + j]);
  }
  __syncthreads();
}

// 1 / x and 1 / sqrt(x) from the hardware estimates (2^-23 relative) and two
// Newton steps each: ~1e-16 relative in ~8 dependent FMAs instead of the ~30
// instruction IEEE sequences.  Used where the result only has to be accurate,
// not correctly rounded (Jacobi rotations: c^2 + s^2 = 1 to rounding is what
// keeps the eigenvectors orthogonal).
__device__ inline double sm_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = fma(fma(-x, y, 1.0), y, y);
  y = fma(fma(-x, y, 1.0), y, y);
  return y;
}
__device__ inline double sm_rsq(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = fma(fma(-hx * y, y, 0.5), y, y);
  y = fma(fma(-hx * y, y, 0.5), y, y);
  return y;
}

// Block-wide sum of a double (all threads get it); scratch >= 4 doubles.
__device__ inline double sm_block_sum(double v, double* scratch) { return block_sum(v, scratch); }

// One-sided (Hestenes) Jacobi on the ROWS of the LOWER-TRIANGULAR matrix A
// (K x K, LDS): plane rotations are applied on the left of A until its rows are
// mutually orthogonal, A' = Vt A.  Then A A^T = Q diag(lam) Q^T with
// Q[i][k] = Vt[k][i], lam[k] = |row k of A'|^2.  8 lanes per row pair, K/2
// independent pairs per round (round-robin tournament), one barrier per round.
// Every round re-reads and re-writes all of A, and the LDS bandwidth of that is
// what a round costs -- so Vt is NOT rotated along (it would double the
// traffic): it is recovered at the end as Vt = A' A0^-1 with one triangular
// solve against the original A (kept in the Vt slot meanwhile).
// A sweep in which every pair was already orthogonal to 1e-8 relative is the
// last one: cyclic Jacobi converges quadratically, the rotations of that sweep
// leave 1e-16, so the confirming sweep is skipped.
// warm (nullable): an orthogonal K x K matrix (row-major, global memory) that
// nearly diagonalises A A^T already -- the Vt of a previous call for a nearby
// A.  The iteration then starts from A <- warm A and needs 1-2 sweeps instead of
// 8-10.  T: scratch [K][KP].
#ifdef SMJ_STAMP
__device__ long long smj_acc[8];
#define SMJ_T(k) { if (threadIdx.x == 0) { const long long tn_ = __builtin_readcyclecounter(); smj_acc[k] += tn_ - smj_t0; smj_t0 = tn_; } }
#else
#define SMJ_T(k)
#endif
// rot2 / done2: squared relative off-diagonal |<p,q>|^2 / (|p|^2 |q|^2) above
// which a pair is rotated / above which another sweep follows.  Results that
// leave in float32 do not need the eigen-decomposition to 1e-13: (1e-18, 1e-10)
// ends one or two sweeps earlier than the float64 setting (1e-26, 1e-16).
__device__ inline void sm_jacobi_rows(double* A, double* Vt, double* lam, int* flag, int K, int KP,
                                      const double* warm, double* T, double rot2 = 1e-26,
                                      double done2 = 1e-16) {
#ifdef SMJ_STAMP
  long long smj_t0 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { for (int k = 0; k < 8; ++k) smj_acc[k] = 0; smj_acc[7] = 0; }
#endif
  double* A0 = Vt;                       // the original A lives in the Vt slot until the end
  for (int e = threadIdx.x; e < K * K; e += SM_BT) {
    const int i = e / K, j = e - i * K;
    A0[i * KP + j] = A[i * KP + j];
    if (warm != nullptr) T[i * KP + j] = warm[e];
  }
  __syncthreads();
  if (warm != nullptr) sm_mm_nn(A, T, A0, K, KP);          // A <- warm A0
  // squared row norms, kept up to date by the rotations (alpha' = alpha - t
  // gamma, beta' = beta + t gamma): a round then needs ONE dot product and one
  // cross-lane reduction instead of three
  __shared__ double nrm[64];
  if (threadIdx.x < K) {
    double s2 = 0;
    for (int k = 0; k < K; ++k) s2 += A[threadIdx.x * KP + k] * A[threadIdx.x * KP + k];
    nrm[threadIdx.x] = s2;
  }
  __syncthreads();
  const int KE = (K + 1) & ~1;          // even player count (K odd: one bye)
  const int npairs = KE / 2;            // <= 32
  const int pair = threadIdx.x >> 3, l8 = threadIdx.x & 7;
  for (int sweep = 0; sweep < 30; ++sweep) {
    if (threadIdx.x == 0) *flag = 0;
    __syncthreads();
    for (int r = 0; r < KE - 1; ++r) {
      int p = 0, q = 0;
      bool on = false;
      if (pair < npairs) {
        if (pair == 0) { p = KE - 1; q = r; }
        else {
          p = r + pair; if (p >= KE - 1) p -= KE - 1;
          q = r - pair; if (q < 0) q += KE - 1;
        }
        on = p < K && q < K;                       // else: bye
        if (!on) { p = 0; q = 0; }
      }
      // this lane's elements k = l8, l8 + 8, ... of both rows stay in registers
      // for the dot products and the rotation (K <= 64: at most 8 each); loads
      // are clamped instead of predicated (no branches on the critical path)
      SMJ_T(0)
      double xa[8], ya[8];
      double ga = 0;
      const double al = nrm[p], be = nrm[q];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = l8 + 8 * i;
        const int kc = k < K ? k : K - 1;
        const double xv = A[p * KP + kc], yv = A[q * KP + kc];
        xa[i] = k < K ? xv : 0.0;
        ya[i] = k < K ? yv : 0.0;
        ga += xa[i] * ya[i];
      }
      SMJ_T(1)
      ga = dpp_sum8(ga);
      SMJ_T(2)
      const double g2 = ga * ga, ab = al * be;
      if (on && g2 > rot2 * ab && ga != 0.0) {
        // t = gamma / (d + sign(d) sqrt(d^2 + gamma^2)), d = (beta - alpha) / 2:
        // the smaller root, |t| <= 1.  The rotation sits on the critical path
        // of every round and float64 operations are slow to depend on here
        // (~20 cycles each): t only steers the convergence, so its square root
        // and reciprocal take ONE Newton step (1e-14); c = 1 / sqrt(1 + t^2)
        // takes two -- c^2 + s^2 = 1 to rounding is what keeps Vt orthogonal.
        const double d = 0.5 * (be - al);
        const double h = fma(d, d, g2);
        double rs = __builtin_amdgcn_rsq(h);
        rs = fma(fma(-0.5 * h * rs, rs, 0.5), rs, rs);
        const double den = fabs(d) + h * rs;
        double it = __builtin_amdgcn_rcp(den);
        it = fma(fma(-den, it, 1.0), it, it);
        const double t = (d >= 0 ? ga : -ga) * it;
        const double c = sm_rsq(fma(t, t, 1.0)), sn = c * t;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int k = l8 + 8 * i;
          if (k < K) {
            A[p * KP + k] = c * xa[i] - sn * ya[i];
            A[q * KP + k] = sn * xa[i] + c * ya[i];
          }
        }
        if (l8 == 0) {
          nrm[p] = al - t * ga;
          nrm[q] = be + t * ga;
          atomicMax(flag, g2 > done2 * ab ? 2 : 1);
        }
      }
      SMJ_T(3)
      __syncthreads();
      SMJ_T(4)
    }
    const int lvl = *flag;
    __syncthreads();
#ifdef SMJ_STAMP
    if (threadIdx.x == 0) smj_acc[7] += 1;
#endif
    if (lvl < 2) break;
  }
  if (threadIdx.x < K) {
    double s = 0;
    for (int k = 0; k < K; ++k) s += A[threadIdx.x * KP + k] * A[threadIdx.x * KP + k];
    lam[threadIdx.x] = s;
  }
  // Vt = A' A0^-1  (Z A0 = A': one solve per row of A')
  for (int e = threadIdx.x; e < K * K; e += SM_BT) {
    const int i = e / K, j = e - i * K;
    T[i * KP + j] = A[i * KP + j];
  }
  __syncthreads();
  sm_trsm_r(T, A0, K, KP);
  for (int e = threadIdx.x; e < K * K; e += SM_BT) {
    const int i = e / K, j = e - i * K;
    Vt[i * KP + j] = T[i * KP + j];
  }
  __syncthreads();
}
