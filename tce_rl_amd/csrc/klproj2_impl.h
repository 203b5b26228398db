// Body of csrc/klproj2.h, compiled once per padded size: the including file
// defines KLP_NS (namespace) and KLP_N (64 or 32).  No include guard on purpose.
namespace KLP_NS {

// N: padded matrix size (64: K <= 64; 32: K <= 32 -- half the block steps, a
// quarter of the work per step, two waves instead of four)
constexpr int N = KLP_N, LOGN = N == 64 ? 6 : 5, P = N + 2, SZ = N * P, BT = 4 * N;
constexpr int NB = N / 4;                  // block steps / elements per thread and image
constexpr int NW = N / 16;                 // waves = 16-row block rows = 16-column tiles
static_assert(N == 64 || N == 32, "klproj2: N");
typedef mfma16_f64x4 d4;

struct Lane {
  int tid, w, x, q;
  __device__ Lane() {
    tid = threadIdx.x;
    w = __builtin_amdgcn_readfirstlane(tid >> 6);
    x = tid & 15;
    q = (tid >> 4) & 3;
  }
};

// C = op(A B^T): wave w forms block row w; lanes read A[16 w + x][k + q], B[16 j + x][k + q].
// f(r, c, v) -> value stored to C[r][c]; C may alias A or B.
template <typename F>
__device__ __forceinline__ void mm_nt(double* C, const double* A, const double* B, const Lane& ln, F f) {
  d4 acc[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j) acc[j] = d4{0, 0, 0, 0};
  const double* ap = A + (16 * ln.w + ln.x) * P + ln.q;
  const double* bp = B + ln.x * P + ln.q;
#pragma unroll
  for (int k0 = 0; k0 < N; k0 += 4) {
    const double a = ap[k0];
#pragma unroll
    for (int j = 0; j < NW; ++j) acc[j] = mfma16(a, bp[16 * j * P + k0], acc[j]);
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NW; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 16 * ln.w + 4 * i + ln.q, c = 16 * j + ln.x;
      C[r * P + c] = f(r, c, acc[j][i]);
    }
  __syncthreads();
}
// C = op(A^T B): contraction index k = kb + s + 8 q (s < 8, kb in {0, 32}); lanes
// read A[k][16 w + x], B[k][16 j + x].
template <typename F>
__device__ __forceinline__ void mm_tn(double* C, const double* A, const double* B, const Lane& ln, F f) {
  d4 acc[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j) acc[j] = d4{0, 0, 0, 0};
  const double* ap = A + 8 * ln.q * P + 16 * ln.w + ln.x;
  const double* bp = B + 8 * ln.q * P + ln.x;
#pragma unroll
  for (int kb = 0; kb < N; kb += 32)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const double a = ap[(kb + s) * P];
#pragma unroll
      for (int j = 0; j < NW; ++j) acc[j] = mfma16(a, bp[(kb + s) * P + 16 * j], acc[j]);
    }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NW; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 16 * ln.w + 4 * i + ln.q, c = 16 * j + ln.x;
      C[r * P + c] = f(r, c, acc[j][i]);
    }
  __syncthreads();
}
struct Ident {
  __device__ double operator()(int, int, double v) const { return v; }
};
struct Tril {
  __device__ double operator()(int r, int c, double v) const { return c <= r ? v : 0.0; }
};

// elementwise over the 64 x 64 image: D[r][c] = f(r, c); barrier behind it
template <typename F>
__device__ __forceinline__ void each(double* D, F f) {
  for (int e = threadIdx.x; e < N * N; e += BT) {
    const int r = e >> LOGN, c = e & (N - 1);
    D[r * P + c] = f(r, c);
  }
  __syncthreads();
}
// sum over the image of f(r, c) (all threads get it)
template <typename F>
__device__ __forceinline__ double total(F f, double* red) {
  double s = 0;
  for (int e = threadIdx.x; e < N * N; e += BT) s += f(e >> LOGN, e & (N - 1));
  return block_sum(s, red);
}
// global [K][K] (row-major, type T) -> image, identity / zero padded
template <typename T>
__device__ __forceinline__ void load(double* D, const T* __restrict__ src, int K, bool tril, bool ident,
                            bool transpose = false) {
  for (int e = threadIdx.x; e < N * N; e += BT) {
    const int r = e >> LOGN, c = e & (N - 1);
    double v = (ident && r == c) ? 1.0 : 0.0;
    if (r < K && c < K && !(tril && (transpose ? r > c : c > r)))
      v = (double)(transpose ? src[c * K + r] : src[r * K + c]);
    D[r * P + c] = v;
  }
  __syncthreads();
}
__device__ __forceinline__ void store_ctx(double* __restrict__ dst, const double* S, int K) {
  for (int e = threadIdx.x; e < K * K; e += BT) {
    const int r = e / K, c = e - r * K;
    dst[e] = S[r * P + c];
  }
}

// 1 / x and 1 / sqrt(x) from the hardware estimates + two Newton steps (~1e-16
// relative, ~8 dependent FMAs; an IEEE division is ~30 instructions on the
// critical path of every block step)
__device__ __forceinline__ double frcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = fma(fma(-x, y, 1.0), y, y);
  y = fma(fma(-x, y, 1.0), y, y);
  return y;
}
__device__ __forceinline__ double frsq(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = fma(fma(-hx * y, y, 0.5), y, y);
  y = fma(fma(-hx * y, y, 0.5), y, y);
  return y;
}
// ---- 4 x 4 helpers in registers (every thread forms them for itself) ----------
// element `idx` of four values by selects (a runtime index into a register array
// would move the array to scratch memory)
__device__ __forceinline__ double pick4(double a0, double a1, double a2, double a3, int idx) {
  const double lo = idx & 1 ? a1 : a0, hi = idx & 1 ? a3 : a2;
  return idx & 2 ? hi : lo;
}
// Pi = P^-1 (Gauss-Jordan without pivoting: SPD or triangular P); returns det P
__device__ __forceinline__ double inv4(const double (&Pm)[4][4], double (&Pi)[4][4]) {
  double a[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) a[r][c] = Pm[r][c];
  double det = 1.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double piv = a[k][k];
    det *= piv;
    const double rp = frcp(piv);
#pragma unroll
    for (int c = 0; c < 4; ++c) a[k][c] *= rp;
    a[k][k] = rp;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (r == k) continue;
      const double f = a[r][k];
#pragma unroll
      for (int c = 0; c < 4; ++c) a[r][c] -= f * a[k][c];
      a[r][k] = -f * rp;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) Pi[r][c] = a[r][c];
  return det;
}
// Ct = (chol(D))^-T for a 4 x 4 SPD block D (lower part used): the matrix that
// turns a row F of the panel below D into the row F Ct of the Cholesky factor;
// Lc = chol(D)
__device__ __forceinline__ void chol4_inv_t(const double (&D)[4][4], double (&Lc)[4][4],
                                            double (&Ct)[4][4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) Lc[r][c] = 0.0;
  double rsd[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double d = D[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= Lc[j][k] * Lc[j][k];
    const double rs = frsq(d);
    rsd[j] = rs;
    Lc[j][j] = d * rs;
#pragma unroll
    for (int i = j + 1; i < 4; ++i) {
      double v = D[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) v -= Lc[i][k] * Lc[j][k];
      Lc[i][j] = v * rs;
    }
  }
  // T = Lc^-1 (lower), Ct = T^T
  double T[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) T[r][c] = 0.0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    T[c][c] = rsd[c];
#pragma unroll
    for (int r = c + 1; r < 4; ++r) {
      double v = 0;
#pragma unroll
      for (int k = c; k < r; ++k) v -= Lc[r][k] * T[k][c];
      T[r][c] = v * rsd[r];
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) Ct[r][c] = T[c][r];
}

// In-place BLOCK Gauss-Jordan inversion of the image (no pivoting: SPD or
// triangular input with pivots away from zero), pivot blocks of 4: the matrix is
// spread over the registers of the 256 threads -- thread t holds row t / 4,
// columns (t % 4) + 4 m, so the 4 pivot columns of block step m are element m of
// the 4 threads of a row -- and there is ONE barrier per block step (16 in all;
// the scalar form measured 1 050 cycles per pivot, almost all of it the barrier
// + LDS round trip + reciprocal chain): the 4 pivot rows (4 x 64) and the 4
// pivot columns (64 x 4) go through two alternating LDS strips, every thread
// inverts the 4 x 4 pivot block for itself.  Block formulas:
//   rows of the block:  A[K][j] <- Pinv A[K][j],  A[K][K] <- Pinv
//   other rows:         G = A[i][K] Pinv;  A[i][j] -= G A[K][j];  A[i][K] <- -G
// Returns log(product of the pivot-block determinants) = log det, in every
// thread (the products are of numbers near 1 where the value is used).
// strips: [2][(4 x 64) + (64 x 4)] doubles.
__device__ __forceinline__ double gj_inverse(double* S, double* strips) {
  const int t = threadIdx.x, i = t >> 2, jb = t & 3;
  double v[NB];
#pragma unroll
  for (int m = 0; m < NB; ++m) v[m] = S[i * P + jb + 4 * m];
  double det = 1.0;
  auto publish = [&](int m, double* st) {          // strips of block step m
    if ((i >> 2) == m) {
#pragma unroll
      for (int mm = 0; mm < NB; ++mm) st[(i & 3) * N + jb + 4 * mm] = v[mm];
    }
  };
  publish(0, strips);
  strips[4 * N + i * 4 + jb] = v[0];
  __syncthreads();
#pragma unroll
  for (int m = 0; m < NB; ++m) {
    const double* rows = strips + (m & 1) * 8 * N;   // [4][N]
    const double* cols = rows + 4 * N;               // [N][4]
    double Pm[4][4], Pi[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) Pm[r][c] = rows[r * N + 4 * m + c];
    det *= inv4(Pm, Pi);
    double F[4], G[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) F[c] = cols[i * 4 + c];
    const bool own = (i >> 2) == m;                // a row of the pivot block
    const int rr = i & 3;
    // G = F Pinv (other rows) / row rr of Pinv (pivot rows: then G R = (Pinv R)[rr])
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double g = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) g += F[k] * Pi[k][c];
      G[c] = own ? pick4(Pi[0][c], Pi[1][c], Pi[2][c], Pi[3][c], rr) : g;
    }
    const double sgn = own ? 1.0 : -1.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) G[k] *= sgn;
#pragma unroll
    for (int mm = 0; mm < NB; ++mm) {
      // (4 columns at a time: without the fence the scheduler loads all 64 row
      // values of the step up front, 128 registers, and the kernel spills)
      if ((mm & 3) == 0) __builtin_amdgcn_sched_barrier(0);
      double acc = own ? 0.0 : v[mm];
#pragma unroll
      for (int k = 0; k < 4; ++k) acc += G[k] * rows[k * N + jb + 4 * mm];
      v[mm] = acc;
    }
    __builtin_amdgcn_sched_barrier(0);
    // (G was negated for the other rows above: their block columns get -F Pinv;
    // a pivot row holds its row of Pinv in G)
    v[m] = pick4(G[0], G[1], G[2], G[3], jb);
    if (m + 1 < NB) {
      double* nst = strips + ((m + 1) & 1) * 8 * N;
      publish(m + 1, nst);
      nst[4 * N + i * 4 + jb] = v[m + 1];
      __syncthreads();
    }
  }
#pragma unroll
  for (int m = 0; m < NB; ++m) S[i * P + jb + 4 * m] = v[m];
  __syncthreads();
  return log(det);
}

// In-place Cholesky factor (lower; the strict upper part is zeroed) of the SPD
// image: block LDL^T elimination with 4 x 4 pivot blocks in the same register
// distribution, ONE barrier per block step.  By symmetry the pivot rows are the
// transposed pivot columns, so only the column strip F (64 x 4) is exchanged:
//   rows of block m:   C[K][K] = chol(D),            D = A[K][K]
//   rows below:        C[i][K] = F_i chol(D)^-T
//   trailing part:     A[i][j] -= F_i D^-1 F_j^T = C[i][K] . C[j][K]
// strips: [2][64 x 4] doubles.
__device__ __forceinline__ void cholesky(double* S, double* strips) {
  const int t = threadIdx.x, i = t >> 2, jb = t & 3;
  double v[NB];
#pragma unroll
  for (int m = 0; m < NB; ++m) v[m] = S[i * P + jb + 4 * m];
  strips[i * 4 + jb] = v[0];
  __syncthreads();
#pragma unroll
  for (int m = 0; m < NB; ++m) {
    const double* cols = strips + (m & 1) * 4 * N;   // F [N][4]
    double D[4][4], Lc[4][4], Ct[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) D[r][c] = cols[(4 * m + r) * 4 + c];
    chol4_inv_t(D, Lc, Ct);
    // Ci = F_i Ct: this row's entries of the factor in the block's columns;
    // Gi = F_i D^-1 = Ci Ct^T: the trailing update is A[i][j] -= Gi . F_j
    double Ci[4], Gi[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double g = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) g += cols[i * 4 + k] * Ct[k][c];
      Ci[c] = g;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double g = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) g += Ci[c] * Ct[k][c];
      Gi[k] = g;
    }
    const int bi = i >> 2;                          // this row's block
#pragma unroll
    for (int mm = 0; mm < NB; ++mm) {
      if ((mm & 3) == 0) __builtin_amdgcn_sched_barrier(0);
      if (mm <= m) continue;                        // columns right of the pivot block only
      const int j = jb + 4 * mm;
      double acc = v[mm];
#pragma unroll
      for (int k = 0; k < 4; ++k) acc -= Gi[k] * cols[j * 4 + k];
      v[mm] = acc;
    }
    __builtin_amdgcn_sched_barrier(0);
    // the block's own columns: chol(D) in its rows, F_i Ct below, zero above
    {
      const int ri = i & 3;
      const double lrow = pick4(pick4(Lc[0][0], Lc[1][0], Lc[2][0], Lc[3][0], ri),
                                pick4(Lc[0][1], Lc[1][1], Lc[2][1], Lc[3][1], ri),
                                pick4(Lc[0][2], Lc[1][2], Lc[2][2], Lc[3][2], ri),
                                pick4(Lc[0][3], Lc[1][3], Lc[2][3], Lc[3][3], ri), jb);
      const double below = pick4(Ci[0], Ci[1], Ci[2], Ci[3], jb);
      v[m] = bi == m ? (ri >= jb ? lrow : 0.0) : (bi > m ? below : 0.0);
    }
    if (m + 1 < NB) {
      strips[((m + 1) & 1) * 4 * N + i * 4 + jb] = v[m + 1];
      __syncthreads();
    }
  }
#pragma unroll
  for (int m = 0; m < NB; ++m) {
    const int j = jb + 4 * m;
    S[i * P + j] = j <= i ? v[m] : 0.0;
  }
  __syncthreads();
}

__host__ __device__ inline int64_t ctx_len(int K) { return 4 * (int64_t)K * K + 8; }

// ---------------------------------------------------------------------------
template <typename real>
__global__ __launch_bounds__(BT) void fwd_kernel(
    const real* __restrict__ L, const real* __restrict__ Lo, int64_t sLo, double eps,
    const real* __restrict__ beta, int entropy_eq, real* __restrict__ projL,
    double* __restrict__ ctx, int K, int warm_start) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* B0 = reinterpret_cast<double*>(smem_raw);
  double* B1 = B0 + SZ;
  double* B2 = B1 + SZ;
  double* B3 = B2 + SZ;
  double* strips = B3 + SZ;                              // [16 N]
  __shared__ double red[NW];
  const Lane ln;
  const int64_t b = blockIdx.x;
  const real* Lb = L + b * (int64_t)K * K;
  const real* Lob = Lo + b * sLo;
  double* cb = ctx + b * ctx_len(K);
  const int64_t KK = (int64_t)K * K;
  double* tail = cb + 4 * KK;
  const bool warm = warm_start != 0;
  const bool have_to = warm && tail[4] == 1.0;
  const double eta_prev = (warm && tail[1] == 1.0) ? tail[0] : 0.0;

  load(B0, Lb, K, true, true);                          // L
  // ---- To^T -> B3
  if (have_to) {
    load(B3, cb, K, false, true, true);                 // To transposed (upper triangular)
  } else {
    load(B2, Lob, K, true, true);
    gj_inverse(B2, strips);                             // To
    each(B2, [&](int r, int c) { return c <= r ? B2[r * P + c] : 0.0; });
    store_ctx(cb, B2, K);
    each(B3, [&](int r, int c) { return B2[c * P + r]; });
  }
  // ---- A = To L (lower) -> B2, M = A A^T -> B3
  mm_tn(B2, B3, B0, ln, Tril());
  store_ctx(cb + KK, B2, K);
  double la = 0, fro = 0;
  {
    double s1 = 0, s2 = 0;
    for (int e = threadIdx.x; e < N * N; e += BT) {
      const int r = e >> LOGN, c = e & (N - 1);
      const double a = B2[r * P + c];
      if (r < K && c < K) s1 += a * a;
      if (r == c && r < K) s2 += log(a);
    }
    fro = block_sum(s1, red);
    la = block_sum(s2, red);
  }
  const double logdetM = 2.0 * la;
  const double kl0 = 0.5 * (fro - (double)K - logdetM);
  const bool active = kl0 > eps;                         // block-uniform
  bool bad = false;                                      // the search for eta failed
  double eta = 0;
  if (active) {
    mm_nt(B3, B2, B2, ln, Ident());                      // M
    // ---- Newton on phi(eta) = h^-1/2 - eps^-1/2, bracketed.  Start: the Newton
    // step of phi from eta = 0, where everything is known without an inversion
    // (h(0) = kl0, h'(0) = -1/2 |M - I|_F^2); the previous call's eta instead
    // when the context is warm and the two are within a factor 4
    double lo = 0.0, hi = -1.0;
    const double se = 1.0 / sqrt(eps);
    {
      const double f2 = total([&](int r, int c) {
        const double d = B3[r * P + c] - (r == c ? 1.0 : 0.0);
        return d * d; }, red);
      const double r0 = 1.0 / sqrt(kl0);
      eta = (se - r0) / (0.25 * r0 * r0 * r0 * f2);
      if (eta_prev > 0.25 * eta && eta_prev < 4.0 * eta) eta = eta_prev;
    }
    int evals = 0;
    bool met = false;                                    // (block-uniform, as eta)
    double h_last = __longlong_as_double(0x7ff8000000000000ll);
    const bool refine = kl0 > 0.5;
    for (int it = 0; it < 60; ++it) {
      ++evals;
      const double e1 = eta + 1.0, re1 = 1.0 / e1;
      each(B1, [&](int r, int c) { return (eta * B3[r * P + c] + (r == c ? 1.0 : 0.0)) * re1; });
      const double logdetN = gj_inverse(B1, strips);     // B1 = W'
      if (refine) {
        // one Newton-Schulz step W' (2 I - N' W'): the elimination has no pivot
        // search and keeps ~cond(N') 1e-16; far outside the trust region
        // (kl0 > 1/2: cond up to 1e8 in the tests' independent draws) that would
        // reach eta -- the step squares the error.  Never taken by a policy update.
        each(B0, [&](int r, int c) { return (eta * B3[r * P + c] + (r == c ? 1.0 : 0.0)) * re1; });
        mm_nt(B0, B0, B1, ln, Ident());                  // N' W'
        mm_tn(B0, B1, B0, ln, Ident());                  // W' (N' W')
        each(B1, [&](int r, int c) { return 2.0 * B1[r * P + c] - B0[r * P + c]; });
      }
      mm_nt(B0, B1, B1, ln, Ident());                    // W'^2
      double t1, t2;
      {
        double s1 = 0, s2 = 0;
        for (int e = threadIdx.x; e < N * N; e += BT) {
          const int r = e >> LOGN, c = e & (N - 1);
          const double d = (r == c ? 1.0 : 0.0) - B3[r * P + c];      // I - M
          s1 += d * B1[r * P + c];
          s2 += d * B0[r * P + c];
        }
        t1 = block_sum(s1, red);                         // <I - M, W'>
        t2 = block_sum(s2, red);                         // <I - M, W'^2>
      }
      const double h = 0.5 * (-t1 * re1 - logdetM + logdetN);
      h_last = h;
      const double hp = 0.5 * ((t1 * re1 - t2 * re1 * re1) / eta - t1 * re1 * re1);
      if (h > eps) lo = eta; else hi = eta;
      const double rh = 1.0 / sqrt(h);
      const double phi = rh - se, dphi = -0.5 * rh * rh * rh * hp;
      double nxt = eta - phi / dphi;
      if (!(nxt > lo) || (hi > 0 && !(nxt < hi))) nxt = hi > 0 ? 0.5 * (lo + hi) : 2.0 * eta;
      // the evaluated eta is kept once the constraint holds to 1e-10 relative or
      // the step falls under 1e-10 relative: h carries ~1e-15 of absolute noise
      // (sums of 64 small terms), i.e. ~1e-11 of eps -- asking for more made the
      // iteration wander (8 evaluations per call where 3 do)
      if (fabs(h - eps) <= 1e-10 * eps || fabs(nxt - eta) <= 1e-10 * eta) {
        met = true;
        break;
      }
      eta = nxt;
    }
    // The search ended without a usable dual variable -- a factor that is not
    // positive definite or not finite makes every comparison above false and
    // eta double 60 times: nothing computed from this eta is a projection.
    // Poison it so that the result is NaN and the caller's NaN check on the
    // losses fires (the reference raises "NAN ... detected" in the same place,
    // mprl/rl/agent/temporal_correlated_agent.py:569-577) instead of a
    // plausible-looking context going into the backward pass.
    // (an exit at the evaluation cap with the constraint met to 1e-6 relative is
    // a slow but sound search -- far outside the trust region the last digits of h
    // are noise --, not a failure)
    bad = !(eta == eta) || !(fabs(eta) < 1e300) ||
          (!met && !(fabs(h_last - eps) <= 1e-6 * eps));
    if (threadIdx.x == 0) tail[5] = (double)evals;     // (diagnostic: evaluations of h)
    // ---- Mt = M W' -> B0, C~ = chol(Mt), L~ = Lo C~
    store_ctx(cb + 2 * KK, B1, K);
    // Mt = (eta + 1) M W.  From eta M W + W = I:  Mt = ((eta + 1) I - W') / eta --
    // no product, so the entries of M (1e4 far outside the trust region) do not
    // multiply the rounding of W'; for eta < 1 (just outside the region, M near
    // I) the difference would cancel instead and the product M W' is the
    // accurate form
    if (eta >= 1.0) {
      const double re = 1.0 / eta;
      each(B0, [&](int r, int c) { return ((r == c ? eta + 1.0 : 0.0) - B1[r * P + c]) * re; });
    } else {
      mm_nt(B0, B3, B1, ln, Ident());
    }
    cholesky(B0, strips);
    store_ctx(cb + 3 * KK, B0, K);
    load(B2, Lob, K, true, true, true);                  // Lo^T
    mm_tn(B2, B2, B0, ln, Tril());                       // Lp
  } else {
    each(B2, [&](int r, int c) { return B0[r * P + c]; });   // Lp = L
  }
  // entropy control: alpha = exp((beta - H)/K) if H < beta (or equality form)
  double alpha = 1.0;
  if (beta != nullptr) {
    double ld = 0;
    for (int i = threadIdx.x; i < K; i += BT) ld += log(B2[i * P + i]);
    const double H = 0.5 * K * (1.0 + 1.8378770664093453) + block_sum(ld, red);
    const double bt = (double)beta[0];
    if (entropy_eq || H < bt) alpha = exp((bt - H) / (double)K);
  }
  real* out = projL + b * KK;
  for (int e = threadIdx.x; e < K * K; e += BT) {
    const int r = e / K, c = e - r * K;
    out[e] = c <= r ? (real)(alpha * B2[r * P + c]) : real(0);
    if (bad) out[e] = (real)__longlong_as_double(0x7ff8000000000000ll);
  }
  if (threadIdx.x == 0) {
    tail[0] = bad ? __longlong_as_double(0x7ff8000000000000ll) : eta;
    tail[1] = active ? 1.0 : 0.0;
    tail[2] = alpha;
    tail[3] = kl0;
    tail[4] = 1.0;
  }
}

template <typename real>
__global__ __launch_bounds__(BT) void bwd_kernel(
    const real* __restrict__ L, const real* __restrict__ Lo, int64_t sLo,
    const real* __restrict__ projL, const double* __restrict__ ctx,
    const real* __restrict__ gproj, real* __restrict__ gL, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* B0 = reinterpret_cast<double*>(smem_raw);
  double* B1 = B0 + SZ;
  double* B2 = B1 + SZ;
  double* B3 = B2 + SZ;
  double* strips = B3 + SZ;
  __shared__ double red[NW];
  const Lane ln;
  const int64_t b = blockIdx.x;
  const int64_t KK = (int64_t)K * K;
  const double* cb = ctx + b * ctx_len(K);
  const double* tail = cb + 4 * KK;
  const double eta = tail[0], alpha = tail[2];
  const bool active = tail[1] != 0.0;
  const real* Lob = Lo + b * sLo;
  real* gb = gL + b * KK;
  (void)L;

  // G (lower) -> B0; entropy scaling: out = alpha(Lp) Lp,
  // dLp = alpha G - (alpha / K) <G, Lp> diag(1 / Lp_ii)
  load(B0, gproj + b * KK, K, true, false);
  if (alpha != 1.0) {
    load(B1, projL + b * KK, K, true, false);
    const double dot = total([&](int r, int c) { return B0[r * P + c] * B1[r * P + c]; }, red) / alpha;
    each(B0, [&](int r, int c) {
      double v = alpha * B0[r * P + c];
      if (r == c && r < K) v -= (alpha / (double)K) * dot / (B1[r * P + r] / alpha);
      return v;
    });
  }
  if (!active) {
    for (int e = threadIdx.x; e < K * K; e += BT) {
      const int r = e / K, c = e - r * K;
      gb[e] = c <= r ? (real)B0[r * P + c] : real(0);
    }
    return;
  }
  const double e1 = eta + 1.0, re1 = 1.0 / e1;
  // Gc = tril(Lo^T G) -> B2
  load(B1, Lob, K, true, true);
  mm_tn(B2, B1, B0, ln, Tril());
  // Phi = tril(C~^T Gc), diagonal halved -> B0
  load(B1, cb + 3 * KK, K, true, true);                  // C~
  mm_tn(B0, B1, B2, ln, [](int r, int c, double v) { return c < r ? v : (c == r ? 0.5 * v : 0.0); });
  // Tc = C~^-1 -> B1; Sbar0 = Tc^T Phi Tc -> B0; Sbar = sym
  gj_inverse(B1, strips);
  each(B1, [&](int r, int c) { return c <= r ? B1[r * P + c] : 0.0; });
  mm_tn(B2, B1, B0, ln, Ident());                        // Tc^T Phi
  each(B3, [&](int r, int c) { return B1[c * P + r]; }); // Tc^T
  mm_nt(B0, B2, B3, ln, Ident());                        // (Tc^T Phi) Tc
  {
    double v[NB];
    for (int m = 0; m < NB; ++m) {
      const int e = threadIdx.x + BT * m, r = e >> LOGN, c = e & (N - 1);
      v[m] = 0.5 * (B0[r * P + c] + B0[c * P + r]);
    }
    __syncthreads();
    for (int m = 0; m < NB; ++m) {
      const int e = threadIdx.x + BT * m, r = e >> LOGN, c = e & (N - 1);
      B0[r * P + c] = v[m];
    }
    __syncthreads();
  }
  // W' -> B1, W'^2 -> B2, M -> B3
  load(B1, cb + 2 * KK, K, false, true);
  mm_nt(B2, B1, B1, ln, Ident());
  load(B3, cb + KK, K, true, true);                      // A
  mm_nt(B3, B3, B3, ln, Ident());                        // M
  // E1 = I - M (B3), E2 = (W - W^2) / eta (B2); scalars of h_eta
  each(B3, [&](int r, int c) { return (r == c ? 1.0 : 0.0) - B3[r * P + c]; });
  const double t1 = total([&](int r, int c) { return B3[r * P + c] * B1[r * P + c]; }, red);   // <I-M, W'>
  each(B2, [&](int r, int c) { return (B1[r * P + c] * re1 - B2[r * P + c] * re1 * re1) / eta; });
  const double t2 = total([&](int r, int c) { return B3[r * P + c] * B2[r * P + c]; }, red);   // <I-M, (W-W^2)/eta>
  const double h_eta = 0.5 * (t2 - t1 * re1 * re1);
  // D = E1 E2 -> B3; c_eta = <Sbar, D>
  mm_nt(B3, B3, B2, ln, Ident());
  const double c_eta = total([&](int r, int c) { return B0[r * P + c] * B3[r * P + c]; }, red);
  const double kappa = c_eta / h_eta;
  // Y = W' Sbar W' -> B0
  mm_nt(B2, B0, B1, ln, Ident());                        // Sbar W'
  mm_tn(B0, B1, B2, ln, Ident());                        // W' (Sbar W')
  // P1 = Y A -> B0;  Z1 = W' A -> B2;  A^-1 -> B3
  load(B3, cb + KK, K, true, true);                      // A
  mm_tn(B0, B0, B3, ln, Ident());
  mm_tn(B2, B1, B3, ln, Ident());
  gj_inverse(B3, strips);                                // A^-1 (lower)
  // Z2 = Z1 - A^-T -> B2; Z3 = W' Z2 -> B2; Abar = 2/(eta+1) (P1 - kappa/2 Z3) -> B0
  {
    double v[NB];
    for (int m = 0; m < NB; ++m) {
      const int e = threadIdx.x + BT * m, r = e >> LOGN, c = e & (N - 1);
      v[m] = B2[r * P + c] - (r <= c ? B3[c * P + r] : 0.0);
    }
    __syncthreads();
    for (int m = 0; m < NB; ++m) {
      const int e = threadIdx.x + BT * m, r = e >> LOGN, c = e & (N - 1);
      B2[r * P + c] = v[m];
    }
    __syncthreads();
  }
  mm_tn(B2, B1, B2, ln, Ident());
  each(B0, [&](int r, int c) { return 2.0 * re1 * (B0[r * P + c] - 0.5 * kappa * B2[r * P + c]); });
  // Lbar = tril(To^T Abar)
  load(B1, cb, K, true, true);                           // To
  mm_tn(B2, B1, B0, ln, Tril());
  for (int e = threadIdx.x; e < K * K; e += BT) {
    const int r = e / K, c = e - r * K;
    gb[e] = c <= r ? (real)B2[r * P + c] : real(0);
  }
}

constexpr size_t LDS_BYTES = (4 * (size_t)SZ + 16 * N) * sizeof(double);

}  // namespace KLP_NS
