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
// X <- L^-1 X   (L lower triangular), one thread per column of X
__device__ inline void sm_trsm_l(double* X, const double* L, int K, int KP) {
  if (threadIdx.x < K) {
    const int c = threadIdx.x;
    for (int r = 0; r < K; ++r) {
      double v = X[r * KP + c];
      for (int k = 0; k < r; ++k) v -= L[r * KP + k] * X[k * KP + c];
      X[r * KP + c] = v / L[r * KP + r];
    }
  }
  __syncthreads();
}
// X <- L^-T X
__device__ inline void sm_trsm_lt(double* X, const double* L, int K, int KP) {
  if (threadIdx.x < K) {
    const int c = threadIdx.x;
    for (int r = K - 1; r >= 0; --r) {
      double v = X[r * KP + c];
      for (int k = r + 1; k < K; ++k) v -= L[k * KP + r] * X[k * KP + c];
      X[r * KP + c] = v / L[r * KP + r];
    }
  }
  __syncthreads();
}
// X <- X L^-1   (solve Z L = X), one thread per row of X
__device__ inline void sm_trsm_r(double* X, const double* L, int K, int KP) {
  if (threadIdx.x < K) {
    const int r = threadIdx.x;
    for (int j = K - 1; j >= 0; --j) {
      double v = X[r * KP + j];
      for (int k = j + 1; k < K; ++k) v -= X[r * KP + k] * L[k * KP + j];
      X[r * KP + j] = v / L[j * KP + j];
    }
  }
  __syncthreads();
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

// One-sided (Hestenes) Jacobi on the ROWS of A (K x K, LDS): rotations J^T are
// applied on the left of A and of Vt (Vt starts as I) until the rows of A are
// mutually orthogonal.  Then A A^T(original) = Q diag(lam) Q^T with
// Q[i][k] = Vt[k][i], lam[k] = |row k of A|^2.  8 lanes per row pair, K/2
// independent pairs per round (round-robin tournament), one barrier per round.
// warm (nullable, with scratch T [K][KP]): an orthogonal K x K matrix (row-major,
// global memory) that nearly diagonalises A A^T already -- the Vt of a previous
// call for a nearby A.  The iteration then starts from Vt = warm, A = warm A and
// needs 2-3 sweeps instead of 8-10.
__device__ inline void sm_jacobi_rows(double* A, double* Vt, double* lam, int* flag, int K, int KP,
                                      const double* warm = nullptr, double* T = nullptr) {
  if (warm != nullptr) {
    for (int e = threadIdx.x; e < K * K; e += SM_BT) Vt[(e / K) * KP + (e % K)] = warm[e];
    __syncthreads();
    for (int e = threadIdx.x; e < K * K; e += SM_BT) {
      const int i = e / K, j = e - i * K;
      double acc = 0;
      for (int k = 0; k < K; ++k) acc += Vt[i * KP + k] * A[k * KP + j];
      T[i * KP + j] = acc;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < K * K; e += SM_BT) {
      const int i = e / K, j = e - i * K;
      A[i * KP + j] = T[i * KP + j];
    }
  } else {
    for (int e = threadIdx.x; e < K * K; e += SM_BT) {
      const int i = e / K, j = e - i * K;
      Vt[i * KP + j] = (i == j) ? 1.0 : 0.0;
    }
  }
  __syncthreads();
  const int KE = (K + 1) & ~1;          // even player count (K odd: one bye)
  const int npairs = KE / 2;            // <= 32
  const int pair = threadIdx.x >> 3, l8 = threadIdx.x & 7;
  for (int sweep = 0; sweep < 30; ++sweep) {
    if (threadIdx.x == 0) *flag = 0;
    __syncthreads();
    for (int r = 0; r < KE - 1; ++r) {
      int p = -1, q = -1;
      if (pair < npairs) {
        if (pair == 0) { p = KE - 1; q = r; }
        else { p = (r + pair) % (KE - 1); q = (r - pair + KE - 1) % (KE - 1); }
        if (p >= K || q >= K) { p = -1; q = -1; }   // bye
      }
      // this lane's elements k = l8, l8 + 8, ... of both rows stay in registers
      // for the dot products and the rotation (K <= 64: at most 8 each)
      double xa[8], ya[8];
      double al = 0, be = 0, ga = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = l8 + 8 * i;
        const bool ok = p >= 0 && k < K;
        xa[i] = ok ? A[p * KP + k] : 0.0;
        ya[i] = ok ? A[q * KP + k] : 0.0;
        al += xa[i] * xa[i]; be += ya[i] * ya[i]; ga += xa[i] * ya[i];
      }
      al = dpp_sum8(al);
      be = dpp_sum8(be);
      ga = dpp_sum8(ga);
      // relative off-diagonal tolerance: eigenvalues to ~1e-13 relative (the
      // bound is 5e-4 .. 5e-2; a tighter test only chases rounding noise)
      if (p >= 0 && ga * ga > 1e-26 * (al * be) && ga != 0.0) {
        // the rotation sits on the critical path of every round (K - 1 rounds
        // per sweep, one barrier each): Newton-refined hardware reciprocals
        const double zeta = (be - al) * sm_rcp(2.0 * ga);
        const double w = fma(zeta, zeta, 1.0);
        const double t = (zeta >= 0 ? 1.0 : -1.0) * sm_rcp(fabs(zeta) + w * sm_rsq(w));
        const double c = sm_rsq(fma(t, t, 1.0)), s = c * t;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int k = l8 + 8 * i;
          if (k < K) {
            A[p * KP + k] = c * xa[i] - s * ya[i];
            A[q * KP + k] = s * xa[i] + c * ya[i];
            const double u = Vt[p * KP + k], v = Vt[q * KP + k];
            Vt[p * KP + k] = c * u - s * v;
            Vt[q * KP + k] = s * u + c * v;
          }
        }
        if (l8 == 0) *flag = 1;
      }
      __syncthreads();
    }
    const int any = *flag;
    __syncthreads();
    if (!any) break;
  }
  if (threadIdx.x < K) {
    double s = 0;
    for (int k = 0; k < K; ++k) s += A[threadIdx.x * KP + k] * A[threadIdx.x * KP + k];
    lam[threadIdx.x] = s;
  }
  __syncthreads();
}
