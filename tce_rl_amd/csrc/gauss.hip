// Gaussian-policy linear algebra for gfx950: Cholesky head, Mahalanobis / KL
// parts, param-space log-prob, and the differentiable KL trust-region
// projection (mean: closed form; covariance: dual multiplier eta + implicit
// gradient).
//
// Reference surface:
//   AbstractGaussianPolicy._vector_to_cholesky   mprl/rl/policy/abstract_policy.py:166-187
//   BlackBoxPolicy.log_prob / maha / ...         mprl/rl/policy/black_box_policy.py:95-224
//   KLProjectionLayer (third-party trust_region_projections + C++ cpp_projection)
//        call sites mprl/rl/projection/__init__.py:18-40,
//        mprl/rl/agent/temporal_correlated_agent.py:530-567
// The covariance projection is ONE workgroup per K x K matrix (K <= 64), all
// in LDS and in double precision like the reference's C++ solver: whiten with
// the old Cholesky factor, one-sided Jacobi eigen-decomposition, scalar root
// find for eta on the eigenvalues, reassemble, Cholesky; the backward pass is
// the closed-form implicit gradient in the eigenbasis.
#include "smallmat.h"
#include "lanevec.h"
#include "klproj2.h"

namespace {

constexpr double LOG_2PI = 1.8378770664093453;
int g_klp_impl = 2;                 // tce_kl_proj_impl: 0 Jacobi, 1 Newton, 2 by size
inline bool klp_newton(int K) { return g_klp_impl == 1 || (g_klp_impl == 2 && K >= 20); }

// ---------------------------------------------------------------------------
// Cholesky head: vec [B, K (+ K(K-1)/2)] -> L [B, K, K]
//   diag = softplus(v[:K]) + min_std (torch threshold 20), off-diagonal filled
//   row-major over tril_indices(K, K, -1).
// ---------------------------------------------------------------------------
template <typename real>
__global__ __launch_bounds__(256) void chol_build_kernel(const real* __restrict__ vec,
                                                         real* __restrict__ L, int64_t B, int K,
                                                         int nvec, real min_std) {
  const int64_t total = B * (int64_t)K * K;
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t b = e / (K * K);
    const int rc = (int)(e - b * K * K);
    const int r = rc / K, c = rc - r * K;
    const real* v = vec + b * nvec;
    real out = 0;
    if (r == c) {
      const real x = v[r];
      out = (x > real(20) ? x : log1p(exp(x))) + min_std;
    } else if (c < r && nvec > K) {
      out = v[K + r * (r - 1) / 2 + c];
    }
    L[e] = out;
  }
}
template <typename real>
__global__ __launch_bounds__(256) void chol_build_bwd_kernel(const real* __restrict__ vec,
                                                             const real* __restrict__ gL,
                                                             real* __restrict__ gvec, int64_t B,
                                                             int K, int nvec) {
  const int64_t total = B * (int64_t)nvec;
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t b = e / nvec;
    const int i = (int)(e - b * nvec);
    const real* g = gL + b * (int64_t)K * K;
    if (i < K) {
      const real x = vec[e];
      const real sig = x > real(20) ? real(1) : real(1) / (real(1) + exp(-x));
      gvec[e] = g[i * K + i] * sig;
    } else {
      // invert i = K + r(r-1)/2 + c
      const int t = i - K;
      int r = (int)((1.0 + sqrt(1.0 + 8.0 * (double)t)) * 0.5);
      while (r * (r - 1) / 2 > t) --r;
      while ((r + 1) * r / 2 <= t) ++r;
      const int c = t - r * (r - 1) / 2;
      gvec[e] = g[r * K + c];
    }
  }
}

// ---------------------------------------------------------------------------
// Per-env vector ops with a lower-triangular L (batch stride sL, 0 = shared):
//   d = L^-1 (x - y),  q = L^-T d
// MODE 0: maha = |d|^2                     bwd: dx = 2 g q (dy = -dx)
// MODE 1: mean projection (KL trust region on the mean)
//           m = 1/2 |d|^2 ; if m > eps: s = sqrt(m/eps),
//           out = (x + (s-1) y) / (s + 1e-16) else out = x
//         bwd: dx = g/s - (g.delta) q / (2 eps s^3)   (active), g (inactive)
// MODE 2: log N(x; y, L L^T)               bwd: dy = g q (dx unused), and
//           dL = g (q d^T - diag(1/L_ii)) per env (lower triangle)
// A per-env L (contextual covariance): one thread per env (vec_env_kernel); a
// shared L (sL = 0, every shipped config): one wave per env, lane = vector
// element (vec_env_shared_kernel, lanevec.h).
// ---------------------------------------------------------------------------
constexpr int VE_MAXK = 64;

template <typename real, int MODE, bool BWD>
__global__ __launch_bounds__(64) void vec_env_kernel(
    const real* __restrict__ x, const real* __restrict__ y, const real* __restrict__ L,
    int64_t sL, real eps, const real* __restrict__ gout, real* __restrict__ out,
    real* __restrict__ gx, real* __restrict__ gLout, int64_t N, int K, int acc,
    real* __restrict__ aux) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // LDS: d[K][64] | q[K][64] (lane = env: conflict free)
  real* dS = reinterpret_cast<real*>(smem_raw);
  real* qS = dS + K * 64;
  const int64_t n = blockIdx.x * 64ll + threadIdx.x;
  if (n >= N) return;
  const real* Ln = L + n * sL;
  real* d = dS + threadIdx.x;          // d[k] -> d[k * 64]
  real quad = 0, logdet = 0;
#pragma unroll 1
  for (int r = 0; r < K; ++r) {
    real v = x[n * K + r] - y[n * K + r];
    for (int k = 0; k < r; ++k) v -= Ln[r * K + k] * d[k * 64];
    const real lrr = Ln[r * K + r];
    v /= lrr;
    d[r * 64] = v;
    quad += v * v;
    if (MODE == 2) logdet += log(lrr);
  }
  if (!BWD) {
    if (MODE == 0) out[n] = quad;
    if (MODE == 2) out[n] = real(-0.5) * quad - logdet - real(0.5 * LOG_2PI) * (real)K;
    if (MODE == 1) {
      if (aux) aux[n] = quad;                       // |L^-1 (x - y)|^2 for the KL diagnostics
      const real m = real(0.5) * quad;
      if (m > eps) {
        const real s = sqrt(m / eps);
        const real om = s - real(1);
        for (int r = 0; r < K; ++r)
          out[n * K + r] = (x[n * K + r] + om * y[n * K + r]) / (real(1) + om + real(1e-16));
      } else {
        for (int r = 0; r < K; ++r) out[n * K + r] = x[n * K + r];
      }
    }
    return;
  }
  // backward: q = L^-T d
  real* q = qS + threadIdx.x;
#pragma unroll 1
  for (int r = K - 1; r >= 0; --r) {
    real v = d[r * 64];
    for (int k = r + 1; k < K; ++k) v -= Ln[k * K + r] * q[k * 64];
    q[r * 64] = v / Ln[r * K + r];
  }
  if (MODE == 0) {
    const real g = gout[n];
    for (int r = 0; r < K; ++r) gx[n * K + r] = real(2) * g * q[r * 64];
  } else if (MODE == 1) {
    const real m = real(0.5) * quad;
    if (m > eps) {
      const real s = sqrt(m / eps);
      real gd = 0;
      for (int r = 0; r < K; ++r) gd += gout[n * K + r] * (x[n * K + r] - y[n * K + r]);
      const real coef = gd / (real(2) * eps * s * s * s);
      for (int r = 0; r < K; ++r) {
        const real o = gout[n * K + r] / s - coef * q[r * 64];
        gx[n * K + r] = acc ? gx[n * K + r] + o : o;
      }
    } else {
      for (int r = 0; r < K; ++r) gx[n * K + r] = acc ? gx[n * K + r] + gout[n * K + r] : gout[n * K + r];
    }
  } else {
    const real g = gout[n];
    for (int r = 0; r < K; ++r) gx[n * K + r] = g * q[r * 64];      // d logp / d mean
    if (gLout) {
      real* gl = gLout + n * (int64_t)K * K;
      for (int r = 0; r < K; ++r)
        for (int c = 0; c < K; ++c) {
          real v = 0;
          if (c <= r) v = g * (q[r * 64] * d[c * 64] - (r == c ? real(1) / Ln[r * K + r] : real(0)));
          gl[r * K + c] = v;
        }
    }
  }
}

// Shared L: wave = env (VS_EPW envs per wave, two at a time), lane = element.
constexpr int VS_BT = 256, VS_EPW = 4, VS_EPB = (VS_BT / 64) * VS_EPW;

template <typename real, int MODE, bool BWD>
__global__ __launch_bounds__(VS_BT) void vec_env_shared_kernel(
    const real* __restrict__ x, const real* __restrict__ y, const real* __restrict__ L,
    real eps, const real* __restrict__ gout, real* __restrict__ out, real* __restrict__ gx,
    real* __restrict__ gLout, int64_t N, int K, int acc, real* __restrict__ aux,
    real* __restrict__ aux2, const real* __restrict__ lp_old = nullptr,
    const real* __restrict__ adv = nullptr, real* __restrict__ logp_out = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  real* Ls = reinterpret_cast<real*>(smem_raw);               // [K][KP]
  const int KP = sm_pitch(K);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < K * K; e += VS_BT) {
    const int i = e / K, j = e - i * K;
    Ls[i * KP + j] = L[e];
  }
  __syncthreads();
  const bool in = lane < K;
  const real lkk = in ? Ls[lane * KP + lane] : real(1);
  const real rdk = in ? real(1) / lkk : real(0);
  real logdet = 0;
  if (MODE == 2) logdet = wave_sum(in ? log(lkk) : real(0));
  const int64_t n0 = (int64_t)blockIdx.x * VS_EPB + wave * VS_EPW;
  for (int i0 = 0; i0 < VS_EPW; i0 += 2) {
    if (n0 + i0 >= N) break;
    int64_t n[2] = {n0 + i0, n0 + i0 + 1};
    const bool ok1 = n[1] < N;
    if (!ok1) n[1] = n[0];                                    // the pair's second env repeats the first
    real xv[2], yv[2], d[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      xv[s] = in ? x[n[s] * K + lane] : real(0);
      yv[s] = in ? y[n[s] * K + lane] : real(0);
      d[s] = xv[s] - yv[s];
    }
    const real* const Lp[2] = {Ls, Ls};
    const real rd[2] = {rdk, rdk};
    if (MODE == 1 && BWD && aux2) {
      // z = L^-1 (x - y) as the forward pass left it (tce_mean_proj_fwd_q_*): the
      // backward is one substitution instead of two
#pragma unroll
      for (int s = 0; s < 2; ++s) d[s] = in ? aux2[n[s] * K + lane] : real(0);
    } else {
      lv_solve_lower<real, 2>(d, Lp, rd, K, KP, lane);
    }
    real quad[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) quad[s] = wave_sum(d[s] * d[s]);
    if (!BWD) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (s == 1 && !ok1) break;
        if (MODE == 0 && lane == 0) out[n[s]] = quad[s];
        if (MODE == 2 && lane == 0)
          out[n[s]] = real(-0.5) * quad[s] - logdet - real(0.5 * LOG_2PI) * (real)K;
        if (MODE == 1 && aux && lane == 0) aux[n[s]] = quad[s];
        if (MODE == 1 && aux2 && in) aux2[n[s] * K + lane] = d[s];
        if (MODE == 1 && in) {
          const real m = real(0.5) * quad[s];
          real o = xv[s];
          if (m > eps) {
            const real om = sqrt(m / eps) - real(1);
            o = (xv[s] + om * yv[s]) / (real(1) + om + real(1e-16));
          }
          out[n[s] * K + lane] = o;
        }
      }
      continue;
    }
    // backward: q = L^-T d (mean projection: only rows outside the trust region
    // use it -- skipped when neither env of the pair is)
    real q[2] = {d[0], d[1]};
    if (MODE != 1 || real(0.5) * quad[0] > eps || real(0.5) * quad[1] > eps)
      lv_solve_lower_t<real, 2>(q, Lp, rd, K, KP, lane);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (s == 1 && !ok1) break;
      if (MODE == 0) {
        if (in) gx[n[s] * K + lane] = real(2) * gout[n[s]] * q[s];
      } else if (MODE == 1) {
        const real go = in ? gout[n[s] * K + lane] : real(0);
        const real m = real(0.5) * quad[s];
        real o = go;
        if (m > eps) {                                        // uniform over the wave
          const real sc = sqrt(m / eps);
          const real gd = wave_sum(go * (xv[s] - yv[s]));
          const real coef = gd / (real(2) * eps * sc * sc * sc);
          o = go / sc - coef * q[s];
        }
        // (acc: the mean projection's backward ADDS to what gx holds -- the other
        // half of d / d mean, written on another stream and waited for by the caller)
        if (in) gx[n[s] * K + lane] = acc ? gx[n[s] * K + lane] + o : o;
      } else {
        real g;
        if (lp_old) {
          // the surrogate's gradient formed here (it needs no sum over the envs):
          // d (-mean(ratio adv)) / d logp_n = -ratio_n adv_n / N, ratio = exp(logp -
          // logp_old); the log-prob is left behind for the loss value
          // (tce_mvn_logprob_bwd_z_sur_*: no forward pass, no gradient kernel)
          const real lp = real(-0.5) * quad[s] - logdet - real(0.5 * LOG_2PI) * (real)K;
          g = -(exp(lp - lp_old[n[s]]) * adv[n[s]]) * (real(1) / (real)N);
          if (lane == 0) logp_out[n[s]] = lp;
        } else {
          g = gout[n[s]];
        }
        if (in) gx[n[s] * K + lane] = g * q[s];               // d logp / d mean
        // (z = L^-1 (x - y) for a caller that sums d logp / d L over the envs as
        // ONE product (g q)^T z instead of N outer products: tce_mvn_logprob_bwd_z_*)
        if (aux2 && in) aux2[n[s] * K + lane] = d[s];
        if (gLout) {
          real* gl = gLout + n[s] * (int64_t)K * K;
          for (int r = 0; r < K; ++r) {
            const real qr = lv_bcast(q[s], r);
            real v = 0;
            if (lane <= r) v = g * (qr * d[s] - (lane == r ? rdk : real(0)));
            if (in) gl[r * K + lane] = v;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// KL covariance part  c(L, Lo) = 1/2 (|Lo^-1 L|_F^2 - K - 2 sum log A_ii),
// A = Lo^-1 L; one workgroup per matrix.  bwd: dL = g (Lo^-T A - diag(1/L_ii)).
// ---------------------------------------------------------------------------
template <typename real, bool BWD>
__global__ __launch_bounds__(SM_BT) void kl_cov_part_kernel(
    const real* __restrict__ L, const real* __restrict__ Lo, int64_t sLo,
    const real* __restrict__ gout, real* __restrict__ out, real* __restrict__ gL, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* A = reinterpret_cast<double*>(smem_raw);
  const int KP = sm_pitch(K);
  double* Los = A + K * KP;
  __shared__ double red[4];
  const int64_t b = blockIdx.x;
  sm_load(A, L + b * (int64_t)K * K, K, KP, true);
  sm_load(Los, Lo + b * sLo, K, KP, true);
  sm_trsm_l(A, Los, K, KP);
  if (!BWD) {
    double loc = 0;
    for (int e = threadIdx.x; e < K * K; e += SM_BT) {
      const int i = e / K, j = e - i * K;
      const double a = A[i * KP + j];
      loc += a * a;
      if (i == j) loc -= 2.0 * log(a);
    }
    const double tot = sm_block_sum(loc, red);
    if (threadIdx.x == 0) out[b] = (real)(0.5 * (tot - (double)K));
  } else {
    sm_trsm_lt(A, Los, K, KP);           // Lo^-T A
    const double g = (double)gout[b];
    const real* Lb = L + b * (int64_t)K * K;
    real* gb = gL + b * (int64_t)K * K;
    for (int e = threadIdx.x; e < K * K; e += SM_BT) {
      const int i = e / K, j = e - i * K;
      double v = 0;
      if (j <= i) v = g * (A[i * KP + j] - (i == j ? 1.0 / (double)Lb[e] : 0.0));
      gb[e] = (real)v;
    }
  }
}

// ---------------------------------------------------------------------------
// KL covariance projection.  ctx (double) per matrix: [Vt K*K | lam K | eta,
// active, alpha, pad].
// ---------------------------------------------------------------------------
#ifdef KLP_STAMP
// diagnostic build (scripts/klproj_stamps.py): cycle stamps per section behind the ctx
__host__ __device__ inline int64_t klp_ctx_len(int K) { return (int64_t)K * K + K + 4 + 16; }
#define KLP_T(k) { __syncthreads(); if (threadIdx.x == 0) { const long long tn_ = __builtin_readcyclecounter(); stamp_[k] = (double)(tn_ - t0_); t0_ = tn_; } }
#define KLP_T0() long long t0_ = __builtin_readcyclecounter(); double* stamp_ = cb + (int64_t)K * K + K + 4;
#else
// (the context also has to hold what the eigen-free kernels of klproj2.h keep)
__host__ __device__ inline int64_t klp_ctx_len(int K) { return 4 * (int64_t)K * K + K + 8; }
#define KLP_T(k)
#define KLP_T0()
#endif

// h(eta) = 1/2 sum (mu - 1 - log mu), mu = (eta+1) lam / (eta lam + 1); one wave.
__device__ inline double klp_h(double eta, double lam, bool live) {
  double t = 0;
  if (live) {
    const double mu = (eta + 1.0) * lam / (eta * lam + 1.0);
    t = mu - 1.0 - log(mu);
  }
  return 0.5 * wave_sum_f64(t);
}

template <typename real>
__global__ __launch_bounds__(SM_BT) void kl_cov_proj_fwd_kernel(
    const real* __restrict__ L, const real* __restrict__ Lo, int64_t sLo, double eps,
    const real* __restrict__ beta, int entropy_eq, real* __restrict__ projL,
    double* __restrict__ ctx, int K, int warm_start) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int KP = sm_pitch(K);
  double* A = reinterpret_cast<double*>(smem_raw);   // A -> rotated -> Y
  double* Vt = A + K * KP;
  double* Los = Vt + K * KP;                          // Lo -> S -> Lp
  double* Tmp = Los + K * KP;                         // warm start scratch
  __shared__ double lam[64];
  __shared__ double red[4];
  __shared__ double s_eta;
  __shared__ int s_flag;
  const int64_t b = blockIdx.x;
  const real* Lb = L + b * (int64_t)K * K;
  double* cb = ctx + b * klp_ctx_len(K);

  KLP_T0()
  sm_load(A, Lb, K, KP, true);
  sm_load(Los, Lo + b * sLo, K, KP, true);
  KLP_T(0)
  sm_trsm_l(A, Los, K, KP);                           // A = Lo^-1 L (lower)
  KLP_T(1)
  double loc = 0;
  for (int e = threadIdx.x; e < K * K; e += SM_BT) {
    const int i = e / K, j = e - i * K;
    const double a = A[i * KP + j];
    loc += a * a;
    if (i == j) loc -= 2.0 * log(a);
  }
  const double kl0 = 0.5 * (sm_block_sum(loc, red) - (double)K);
  const bool active = kl0 > eps;                      // block-uniform
  double eta = 0;
  if (active) {
    // warm start: ctx still holds the eigenvectors of the previous call (its
    // "active" slot says whether it wrote any)
    const bool warm = warm_start && cb[(int64_t)K * K + K + 1] == 1.0;
    KLP_T(2)
    // float32 results do not need the decomposition beyond their own rounding
    sm_jacobi_rows(A, Vt, lam, &s_flag, K, KP, warm ? cb : nullptr, Tmp,
                   sizeof(real) == 4 ? 1e-18 : 1e-26, sizeof(real) == 4 ? 1e-10 : 1e-16);
    KLP_T(3)
    if (threadIdx.x < 64) {
      const bool live = threadIdx.x < K;
      const double lm = live ? lam[threadIdx.x] : 1.0;
      // h is convex and decreasing in eta, h(0) > eps: Newton from the left
      // converges monotonically; bracket kept for safety, bisection fallback
      double lo = 0.0, hi = 1.0;
      for (int i = 0; i < 200 && klp_h(hi, lm, live) > eps; ++i) { lo = hi; hi *= 2.0; }
      double eta_n = lo;
      for (int i = 0; i < 60; ++i) {
        double t = 0, dt = 0;
        if (live) {
          const double w = 1.0 / (eta_n * lm + 1.0);
          const double mu = (eta_n + 1.0) * lm * w;
          t = mu - 1.0 - log(mu);
          dt = (1.0 - 1.0 / mu) * lm * (1.0 - lm) * w * w;
        }
        const double hv = 0.5 * wave_sum_f64(t) - eps;
        const double dh = 0.5 * wave_sum_f64(dt);
        if (hv > 0) lo = eta_n; else hi = eta_n;
        double nxt = eta_n - hv / dh;
        if (!(nxt > lo && nxt < hi)) nxt = 0.5 * (lo + hi);
        if (fabs(nxt - eta_n) <= 1e-15 * fabs(nxt)) { eta_n = nxt; break; }
        eta_n = nxt;
      }
      if (threadIdx.x == 0) s_eta = eta_n;
    }
    __syncthreads();
    eta = s_eta;
    KLP_T(4)
    // ctx: Vt, lam
    for (int e = threadIdx.x; e < K * K; e += SM_BT) cb[e] = Vt[(e / K) * KP + (e % K)];
    if (threadIdx.x < K) cb[(int64_t)K * K + threadIdx.x] = lam[threadIdx.x];
    // Y[r][k] = sum_{i<=r} Lo[r][i] Vt[k][i] sqrt(mu_k)      -> A
    for (int e = threadIdx.x; e < K * K; e += SM_BT) {
      const int r = e / K, k = e - r * K;
      const double mu = (eta + 1.0) * lam[k] / (eta * lam[k] + 1.0);
      double acc = 0;
      for (int i = 0; i <= r; ++i) acc += Los[r * KP + i] * Vt[k * KP + i];
      A[r * KP + k] = acc * sqrt(mu);
    }
    __syncthreads();
    KLP_T(5)
    sm_mm_nt(Los, A, A, K, KP);                       // S = Y Y^T
    KLP_T(6)
    sm_cholesky(Los, K, KP);                          // Lp
    KLP_T(7)
  } else {
    sm_load(Los, Lb, K, KP, true);                    // Lp = L
  }
  // entropy control: alpha = exp((beta - H)/K) if H < beta (or equality form)
  double alpha = 1.0;
  if (beta != nullptr) {
    double ld = 0;
    for (int i = threadIdx.x; i < K; i += SM_BT) ld += log(Los[i * KP + i]);
    const double H = 0.5 * K * (1.0 + LOG_2PI) + sm_block_sum(ld, red);
    const double bt = (double)beta[0];
    if (entropy_eq || H < bt) alpha = exp((bt - H) / (double)K);
  }
  sm_store(projL + b * (int64_t)K * K, Los, K, KP, true, alpha);
  KLP_T(8)
#ifdef SMJ_STAMP
  if (threadIdx.x == 0 && active)
    for (int k = 0; k < 6; ++k) cb[(int64_t)K * K + K + 4 + 9 + k] = (double)smj_acc[k == 5 ? 7 : k];
#endif
  if (threadIdx.x == 0) {
    double* tail = cb + (int64_t)K * K + K;
    tail[0] = eta;
    tail[1] = active ? 1.0 : 0.0;
    tail[2] = alpha;
    tail[3] = kl0;
  }
}

template <typename real>
__global__ __launch_bounds__(SM_BT) void kl_cov_proj_bwd_kernel(
    const real* __restrict__ L, const real* __restrict__ Lo, int64_t sLo,
    const real* __restrict__ projL, const double* __restrict__ ctx,
    const real* __restrict__ gproj, real* __restrict__ gL, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int KP = sm_pitch(K);
  double* M0 = reinterpret_cast<double*>(smem_raw);
  double* M1 = M0 + K * KP;
  double* M2 = M1 + K * KP;
  double* M3 = M2 + K * KP;
  __shared__ double lam[64], wv[64], mup[64], hB[64];
  __shared__ double red[4];
  const int64_t b = blockIdx.x;
  const double* cb = ctx + b * klp_ctx_len(K);
  const double* tail = cb + (int64_t)K * K + K;
  const double eta = tail[0], alpha = tail[2];
  const bool active = tail[1] != 0.0;
  const real* Lb = L + b * (int64_t)K * K;
  real* gb = gL + b * (int64_t)K * K;

  // Lp = projL / alpha (M0), G (M1)
  sm_load(M0, projL + b * (int64_t)K * K, K, KP, true);
  sm_load(M1, gproj + b * (int64_t)K * K, K, KP, true);
  if (alpha != 1.0) {
    // out = alpha(Lp) Lp: dLp = alpha G - (alpha/K) <G, Lp> diag(1/Lp_ii)
    double loc = 0;
    for (int e = threadIdx.x; e < K * K; e += SM_BT) {
      const int i = e / K, j = e - i * K;
      M0[i * KP + j] /= alpha;
      loc += M1[i * KP + j] * M0[i * KP + j];
    }
    const double dot = sm_block_sum(loc, red);       // <G, Lp>
    for (int e = threadIdx.x; e < K * K; e += SM_BT) {
      const int i = e / K, j = e - i * K;
      double v = alpha * M1[i * KP + j];
      if (i == j) v -= (alpha / (double)K) * dot / M0[i * KP + i];
      M1[i * KP + j] = v;
    }
    __syncthreads();
  }
  if (!active) {
    sm_store(gb, M1, K, KP, true, 1.0);
    return;
  }
  // Cholesky backward: Sbar = 1/2 (Z + Z^T), Z = Lp^-T Phi(Lp^T G) Lp^-1
  sm_mm_tn(M2, M0, M1, K, KP);                        // Lp^T G
  for (int e = threadIdx.x; e < K * K; e += SM_BT) {
    const int i = e / K, j = e - i * K;
    if (j > i) M2[i * KP + j] = 0;
    else if (j == i) M2[i * KP + j] *= 0.5;
  }
  __syncthreads();
  sm_trsm_lt(M2, M0, K, KP);                          // Lp^-T Phi
  sm_trsm_r(M2, M0, K, KP);                           // ... Lp^-1  = Z
  for (int e = threadIdx.x; e < K * K; e += SM_BT) {
    const int i = e / K, j = e - i * K;
    M1[i * KP + j] = 0.5 * (M2[i * KP + j] + M2[j * KP + i]);   // Sbar
  }
  __syncthreads();
  // Gw = Lo^T Sbar Lo
  sm_load(M0, Lo + b * sLo, K, KP, true);             // Lo
  sm_mm_nn(M2, M1, M0, K, KP);                        // Sbar Lo
  sm_mm_tn(M1, M0, M2, K, KP);                        // Lo^T (Sbar Lo) = Gw
  // Ghat = Vt Gw Vt^T
  for (int e = threadIdx.x; e < K * K; e += SM_BT) M3[(e / K) * KP + (e % K)] = cb[e];   // Vt
  if (threadIdx.x < K) {
    const double lm = cb[(int64_t)K * K + threadIdx.x];
    lam[threadIdx.x] = lm;
    const double w = 1.0 / (eta * lm + 1.0);
    wv[threadIdx.x] = w;
    const double mu = (eta + 1.0) * lm * w;
    mup[threadIdx.x] = lm * (1.0 - lm) * w * w;
    hB[threadIdx.x] = 0.5 * (eta + 1.0) * (1.0 - 1.0 / mu) * w * w;
  }
  __syncthreads();
  sm_mm_nn(M2, M3, M1, K, KP);                        // Vt Gw
  sm_mm_nt(M1, M2, M3, K, KP);                        // (Vt Gw) Vt^T = Ghat
  double l1 = 0, l2 = 0;
  for (int i = threadIdx.x; i < K; i += SM_BT) {
    const double mu = (eta + 1.0) * lam[i] * wv[i];
    l1 += (1.0 - 1.0 / mu) * mup[i];                  // 2 h_eta
    l2 += M1[i * KP + i] * mup[i];                    // c_eta
  }
  const double h_eta = 0.5 * sm_block_sum(l1, red);
  const double c_eta = sm_block_sum(l2, red);
  const double ratio = c_eta / h_eta;
  for (int e = threadIdx.x; e < K * K; e += SM_BT) {
    const int i = e / K, j = e - i * K;
    double v = (eta + 1.0) * wv[i] * M1[i * KP + j] * wv[j];
    if (i == j) v -= ratio * hB[i];
    M1[i * KP + j] = v;                               // Bhat
  }
  __syncthreads();
  // Bbar = Vt^T Bhat Vt
  sm_mm_nn(M2, M1, M3, K, KP);                        // Bhat Vt
  sm_mm_tn(M1, M3, M2, K, KP);                        // Vt^T (Bhat Vt) = Bbar
  // A = Lo^-1 L ; Abar = (Bbar + Bbar^T) A ; Lbar = tril(Lo^-T Abar)
  sm_load(M2, Lb, K, KP, true);
  sm_trsm_l(M2, M0, K, KP);                           // A
  for (int e = threadIdx.x; e < K * K; e += SM_BT) {
    const int i = e / K, j = e - i * K;
    double acc = 0;
    for (int k = 0; k < K; ++k) acc += (M1[i * KP + k] + M1[k * KP + i]) * M2[k * KP + j];
    M3[i * KP + j] = acc;
  }
  __syncthreads();
  sm_trsm_lt(M3, M0, K, KP);
  sm_store(gb, M3, K, KP, true, 1.0);
}

template <typename F>
int set_lds(F kern, size_t lds) {
  if (lds > 48 * 1024)
    tce_lds_limit(reinterpret_cast<const void*>(kern), (size_t)(lds));
  return 0;
}


template <typename real, int MODE, bool BWD>
int vec_env_launch_mb(const real* x, const real* y, const real* L, int64_t sL, real eps,
                      const real* gout, real* out, real* gx, real* gL, int64_t N, int K,
                      hipStream_t st, int acc, real* aux, real* aux2,
                      const real* lp_old = nullptr, const real* adv = nullptr,
                      real* logp_out = nullptr) {
  if (sL == 0) {
    const size_t lds = (size_t)K * sm_pitch(K) * sizeof(real);
    set_lds(vec_env_shared_kernel<real, MODE, BWD>, lds);
    hipLaunchKernelGGL((vec_env_shared_kernel<real, MODE, BWD>),
                       dim3((unsigned)ceil_div(N, VS_EPB)), dim3(VS_BT), lds, st, x, y, L, eps,
                       gout, out, gx, gL, N, K, acc, aux, aux2, lp_old, adv, logp_out);
  } else {
    const size_t lds = 2 * (size_t)K * 64 * sizeof(real);
    set_lds(vec_env_kernel<real, MODE, BWD>, lds);
    hipLaunchKernelGGL((vec_env_kernel<real, MODE, BWD>), dim3((unsigned)ceil_div(N, 64)),
                       dim3(64), lds, st, x, y, L, sL, eps, gout, out, gx, gL, N, K, acc, aux);
  }
  TCE_LAUNCH_CHECK();
  return 0;
}

template <typename real>
int vec_env_launch(int mode, int bwd, const real* x, const real* y, const real* L, int64_t sL,
                   real eps, const real* gout, real* out, real* gx, real* gL, int64_t N, int K,
                   hipStream_t st, int acc = 0, real* aux = nullptr, real* aux2 = nullptr) {
  // (aux2, the stored z of the mean projection: shared factor only)
  if (sL != 0) aux2 = nullptr;
#define VE_CASE(M, B)                                                                       \
  if (mode == M && (bwd != 0) == B)                                                         \
    return vec_env_launch_mb<real, M, B>(x, y, L, sL, eps, gout, out, gx, gL, N, K, st, acc, \
                                         aux, aux2);
  VE_CASE(0, false) VE_CASE(0, true) VE_CASE(1, false) VE_CASE(1, true) VE_CASE(2, false)
  VE_CASE(2, true)
#undef VE_CASE
  tce_set_error("vec_env: unknown mode");
  return 1;
}

}  // namespace

extern "C" {

int64_t tce_kl_cov_proj_ctx_len(int K) { return klp_ctx_len(K); }

// 2 (default): by size -- the eigen-decomposition-free kernels of klproj2.h for
// K >= 20 (padded to 32 x 32 on two waves for K <= 32, to 64 x 64 on four
// above: at K 24, forward / backward per call incl. a synchronisation, 48 - 55 /
// 35 us against 97 - 112 / 51 us for the warm-started Jacobi kernels; at K 63
// 0.15 / 0.16 ms against 0.41 / 0.31 in the C3 step), the Jacobi kernels above
// otherwise; 1 / 0: force one form.  A context written by one form must not be
// read by the other (forward and backward of one evaluation, and a warm start,
// use one form).
int tce_kl_proj_impl(int impl) {
  g_klp_impl = impl < 0 || impl > 2 ? 2 : impl;
  return 0;
}

#define DEFINE_GAUSS(SFX, REAL)                                                   \
  int tce_chol_build_fwd_##SFX(const REAL* vec, REAL* L, int64_t B, int K,        \
                               int nvec, REAL min_std, void* stream) {            \
    TCE_CHECK_ARG(vec && L && B > 0 && K > 0 &&                                   \
                      (nvec == K || nvec == K + K * (K - 1) / 2),                 \
                  "chol_build: bad arguments");                                   \
    const int64_t nb = tmin<int64_t>(ceil_div(B * K * K, 256), 4096);             \
    hipLaunchKernelGGL(chol_build_kernel<REAL>, dim3((unsigned)nb), dim3(256), 0, \
                       (hipStream_t)stream, vec, L, B, K, nvec, min_std);         \
    TCE_LAUNCH_CHECK();                                                           \
    return 0;                                                                     \
  }                                                                               \
  int tce_chol_build_bwd_##SFX(const REAL* vec, const REAL* grad_L,               \
                               REAL* grad_vec, int64_t B, int K, int nvec,        \
                               void* stream) {                                    \
    TCE_CHECK_ARG(vec && grad_L && grad_vec && B > 0 && K > 0,                    \
                  "chol_build_bwd: bad arguments");                               \
    const int64_t nb = tmin<int64_t>(ceil_div(B * nvec, 256), 4096);              \
    hipLaunchKernelGGL(chol_build_bwd_kernel<REAL>, dim3((unsigned)nb),           \
                       dim3(256), 0, (hipStream_t)stream, vec, grad_L, grad_vec,  \
                       B, K, nvec);                                               \
    TCE_LAUNCH_CHECK();                                                           \
    return 0;                                                                     \
  }                                                                               \
  /* mode 0 maha, 1 mean projection, 2 log-prob; bwd != 0: backward */            \
  int tce_vec_env_##SFX(int mode, int bwd, const REAL* x, const REAL* y,          \
                        const REAL* L, int64_t L_stride, REAL eps,                \
                        const REAL* grad_out, REAL* out, REAL* grad_x,            \
                        REAL* grad_L, int64_t N, int K, void* stream) {           \
    TCE_CHECK_ARG(x && y && L && N > 0 && K > 0 && K <= VE_MAXK,                  \
                  "vec_env: bad arguments (K <= 64)");                            \
    TCE_CHECK_ARG(bwd ? (grad_out && grad_x) : (out != nullptr),                  \
                  "vec_env: null output");                                        \
    return vec_env_launch<REAL>(mode, bwd, x, y, L, L_stride, eps, grad_out, out,   \
                                grad_x, grad_L, N, K, (hipStream_t)stream);         \
  }                                                                               \
  /* forward of mode 1 that also stores |L^-1 (x - y)|^2 per env */               \
  int tce_mean_proj_fwd_q_##SFX(const REAL* x, const REAL* y, const REAL* L,      \
                                int64_t L_stride, REAL eps, REAL* out,            \
                                REAL* quad_out, REAL* z_out, int64_t N, int K,    \
                                void* stream) {                                   \
    TCE_CHECK_ARG(x && y && L && out && quad_out && N > 0 && K > 0 &&             \
                      K <= VE_MAXK && (z_out == nullptr || L_stride == 0),        \
                  "mean_proj_fwd_q: bad arguments (K <= 64; z_out: shared L)");   \
    return vec_env_launch<REAL>(1, 0, x, y, L, L_stride, eps, nullptr, out,       \
                                nullptr, nullptr, N, K, (hipStream_t)stream, 0,   \
                                quad_out, z_out);                                 \
  }                                                                               \
  /* backward of mode 2 (log-prob) for a SHARED factor that leaves z = L^-1 (x -  \
     y) behind instead of the per-env d / d L */                                  \
  int tce_mvn_logprob_bwd_z_##SFX(const REAL* x, const REAL* y, const REAL* L,    \
                                  const REAL* grad_out, REAL* grad_mean,          \
                                  REAL* z_out, int64_t N, int K, void* stream) {  \
    TCE_CHECK_ARG(x && y && L && grad_out && grad_mean && z_out && N > 0 &&       \
                      K > 0 && K <= VE_MAXK,                                      \
                  "mvn_logprob_bwd_z: bad arguments (K <= 64)");                  \
    return vec_env_launch<REAL>(2, 1, x, y, L, 0, REAL(0), grad_out, nullptr,     \
                                grad_mean, nullptr, N, K, (hipStream_t)stream, 0, \
                                nullptr, z_out);                                  \
  }                                                                               \
  /* the same with the surrogate's gradient formed inside: grad_out[n] =         \
     -exp(logp_n - logp_old[n]) adv[n] / N; logp_out [N] receives the log-probs   \
     (no forward pass, no tce_surrogate_* gradient) */                            \
  int tce_mvn_logprob_bwd_z_sur_##SFX(                                            \
      const REAL* x, const REAL* y, const REAL* L, const REAL* logp_old,          \
      const REAL* adv, REAL* grad_mean, REAL* z_out, REAL* logp_out, int64_t N,   \
      int K, void* stream) {                                                      \
    TCE_CHECK_ARG(x && y && L && logp_old && adv && grad_mean && z_out &&         \
                      logp_out && N > 0 && K > 0 && K <= VE_MAXK,                 \
                  "mvn_logprob_bwd_z_sur: bad arguments (K <= 64)");              \
    return vec_env_launch_mb<REAL, 2, true>(                                      \
        x, y, L, 0, REAL(0), nullptr, nullptr, grad_mean, nullptr, N, K,          \
        (hipStream_t)stream, 0, nullptr, z_out, logp_old, adv, logp_out);         \
  }                                                                               \
  /* backward of mode 1 (mean projection) that ADDS to grad_x */                  \
  int tce_mean_proj_bwd_acc_##SFX(const REAL* x, const REAL* y, const REAL* L,    \
                                  int64_t L_stride, REAL eps,                     \
                                  const REAL* grad_out, const REAL* z,            \
                                  REAL* grad_x, int64_t N, int K, void* stream) { \
    TCE_CHECK_ARG(x && y && L && grad_out && grad_x && N > 0 && K > 0 &&          \
                      K <= VE_MAXK && (z == nullptr || L_stride == 0),            \
                  "mean_proj_bwd_acc: bad arguments (K <= 64; z: shared L)");     \
    return vec_env_launch<REAL>(1, 1, x, y, L, L_stride, eps, grad_out, nullptr,  \
                                grad_x, nullptr, N, K, (hipStream_t)stream, 1,    \
                                nullptr, const_cast<REAL*>(z));                   \
  }                                                                               \
  int tce_kl_cov_part_##SFX(int bwd, const REAL* L, const REAL* L_old,            \
                            int64_t L_old_stride, const REAL* grad_out,           \
                            REAL* out, REAL* grad_L, int64_t B, int K,            \
                            void* stream) {                                       \
    TCE_CHECK_ARG(L && L_old && B > 0 && K > 0 && K <= 64,                        \
                  "kl_cov_part: bad arguments (K <= 64)");                        \
    const size_t lds = 2 * (size_t)K * sm_pitch(K) * sizeof(double);                  \
    if (bwd) {                                                                    \
      TCE_CHECK_ARG(grad_out && grad_L, "kl_cov_part: null gradient buffers");    \
      set_lds(kl_cov_part_kernel<REAL, true>, lds);                               \
      hipLaunchKernelGGL((kl_cov_part_kernel<REAL, true>), dim3((unsigned)B),     \
                         dim3(SM_BT), lds, (hipStream_t)stream, L, L_old,         \
                         L_old_stride, grad_out, out, grad_L, K);                 \
    } else {                                                                      \
      TCE_CHECK_ARG(out != nullptr, "kl_cov_part: null output");                  \
      set_lds(kl_cov_part_kernel<REAL, false>, lds);                              \
      hipLaunchKernelGGL((kl_cov_part_kernel<REAL, false>), dim3((unsigned)B),    \
                         dim3(SM_BT), lds, (hipStream_t)stream, L, L_old,         \
                         L_old_stride, grad_out, out, grad_L, K);                 \
    }                                                                             \
    TCE_LAUNCH_CHECK();                                                           \
    return 0;                                                                     \
  }                                                                               \
  int tce_kl_cov_proj_fwd_##SFX(const REAL* L, const REAL* L_old,                 \
                                int64_t L_old_stride, double eps_cov,             \
                                const REAL* beta, int entropy_eq, REAL* proj_L,   \
                                double* ctx, int64_t B, int K, int warm_start,    \
                                void* stream) {                                   \
    TCE_CHECK_ARG(L && L_old && proj_L && ctx && B > 0 && K > 0 && K <= 64,       \
                  "kl_cov_proj: bad arguments (K <= 64)");                        \
    if (klp_newton(K) && K <= 32) {                                               \
      set_lds(klp2s::fwd_kernel<REAL>, klp2s::LDS_BYTES);                         \
      hipLaunchKernelGGL(klp2s::fwd_kernel<REAL>, dim3((unsigned)B),              \
                         dim3(klp2s::BT), klp2s::LDS_BYTES, (hipStream_t)stream,  \
                         L, L_old, L_old_stride, eps_cov, beta, entropy_eq,       \
                         proj_L, ctx, K, warm_start);                             \
      TCE_LAUNCH_CHECK();                                                         \
      return 0;                                                                   \
    }                                                                             \
    if (klp_newton(K)) {                                                          \
      set_lds(klp2::fwd_kernel<REAL>, klp2::LDS_BYTES);                           \
      hipLaunchKernelGGL(klp2::fwd_kernel<REAL>, dim3((unsigned)B),               \
                         dim3(klp2::BT), klp2::LDS_BYTES, (hipStream_t)stream, L, \
                         L_old, L_old_stride, eps_cov, beta, entropy_eq, proj_L,  \
                         ctx, K, warm_start);                                     \
      TCE_LAUNCH_CHECK();                                                         \
      return 0;                                                                   \
    }                                                                             \
    const size_t lds = 4 * (size_t)K * sm_pitch(K) * sizeof(double);                  \
    set_lds(kl_cov_proj_fwd_kernel<REAL>, lds);                                   \
    hipLaunchKernelGGL(kl_cov_proj_fwd_kernel<REAL>, dim3((unsigned)B),           \
                       dim3(SM_BT), lds, (hipStream_t)stream, L, L_old,           \
                       L_old_stride, eps_cov, beta, entropy_eq, proj_L, ctx, K,   \
                       warm_start);                                               \
    TCE_LAUNCH_CHECK();                                                           \
    return 0;                                                                     \
  }                                                                               \
  int tce_kl_cov_proj_bwd_##SFX(const REAL* L, const REAL* L_old,                 \
                                int64_t L_old_stride, const REAL* proj_L,         \
                                const double* ctx, const REAL* grad_proj,         \
                                REAL* grad_L, int64_t B, int K, void* stream) {   \
    TCE_CHECK_ARG(L && L_old && proj_L && ctx && grad_proj && grad_L && B > 0 &&  \
                      K > 0 && K <= 64,                                           \
                  "kl_cov_proj_bwd: bad arguments (K <= 64)");                    \
    if (klp_newton(K) && K <= 32) {                                               \
      set_lds(klp2s::bwd_kernel<REAL>, klp2s::LDS_BYTES);                         \
      hipLaunchKernelGGL(klp2s::bwd_kernel<REAL>, dim3((unsigned)B),              \
                         dim3(klp2s::BT), klp2s::LDS_BYTES, (hipStream_t)stream,  \
                         L, L_old, L_old_stride, proj_L, ctx, grad_proj, grad_L,  \
                         K);                                                      \
      TCE_LAUNCH_CHECK();                                                         \
      return 0;                                                                   \
    }                                                                             \
    if (klp_newton(K)) {                                                          \
      set_lds(klp2::bwd_kernel<REAL>, klp2::LDS_BYTES);                           \
      hipLaunchKernelGGL(klp2::bwd_kernel<REAL>, dim3((unsigned)B),               \
                         dim3(klp2::BT), klp2::LDS_BYTES, (hipStream_t)stream, L, \
                         L_old, L_old_stride, proj_L, ctx, grad_proj, grad_L, K); \
      TCE_LAUNCH_CHECK();                                                         \
      return 0;                                                                   \
    }                                                                             \
    const size_t lds = 4 * (size_t)K * sm_pitch(K) * sizeof(double);                  \
    set_lds(kl_cov_proj_bwd_kernel<REAL>, lds);                                   \
    hipLaunchKernelGGL(kl_cov_proj_bwd_kernel<REAL>, dim3((unsigned)B),           \
                       dim3(SM_BT), lds, (hipStream_t)stream, L, L_old,           \
                       L_old_stride, proj_L, ctx, grad_proj, grad_L, K);          \
    TCE_LAUNCH_CHECK();                                                           \
    return 0;                                                                     \
  }

DEFINE_GAUSS(f32, float)
DEFINE_GAUSS(f64, double)

}  // extern "C"
