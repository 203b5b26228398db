// Loss-side pieces of one TCE policy epoch with a shared (non-contextual)
// covariance, fused so that the epoch is a short chain of kernels:
//   surrogate     : -mean(exp(lp_new - lp_old) * A) and its gradient w.r.t. lp_new
//                   (mprl/rl/agent/temporal_correlated_agent.py:718-739)
//   kl_shared     : the 12 gaussian_kl_details means of kl_old_new_proj (:641-686),
//                   the entropy of the projected policy (:741-745), the trust
//                   region loss of the projection layer (:562-567, third-party
//                   get_trust_region_loss) and its gradients w.r.t. the new mean /
//                   Cholesky factor.
// With one covariance for all envs only the Mahalanobis terms depend on the
// env: thread = env for those (triangular solves against factors held in LDS),
// one workgroup for the K x K parts (fp64 in LDS, smallmat.h).
#include "smallmat.h"

namespace {

// out[0] = -mean(ratio * adv), out[1] = mean(ratio); grad[i] = -ratio_i adv_i / M.
// Grid of blocks with per-block partial sums; the block that finishes last
// adds them in index order (deterministic) and re-arms the ticket.
constexpr int SUR_BT = 256;
constexpr int SUR_MAX_BLOCKS = 128;

template <typename real>
__global__ __launch_bounds__(SUR_BT) void surrogate_kernel(const real* __restrict__ lp_new,
                                                           const real* __restrict__ lp_old,
                                                           const real* __restrict__ adv, int64_t M,
                                                           real* __restrict__ out,
                                                           real* __restrict__ grad,
                                                           double* __restrict__ partials,
                                                           unsigned* __restrict__ ticket) {
  __shared__ double red[4];
  __shared__ bool last;
  double s = 0, sr = 0;
  const real inv = real(1) / (real)M;
  for (int64_t i = (int64_t)blockIdx.x * SUR_BT + threadIdx.x; i < M;
       i += (int64_t)gridDim.x * SUR_BT) {
    const real ratio = exp(lp_new[i] - lp_old[i]);
    const real ra = ratio * adv[i];
    s += (double)ra;
    sr += (double)ratio;
    if (grad) grad[i] = -ra * inv;
  }
  s = block_sum(s, red);
  sr = block_sum(sr, red);
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x] = s;
    partials[2 * blockIdx.x + 1] = sr;
    __threadfence();
    last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (last && threadIdx.x == 0) {
    __threadfence();
    double ts = 0, tr = 0;
    for (unsigned b = 0; b < gridDim.x; ++b) {
      ts += __builtin_nontemporal_load(partials + 2 * b);
      tr += __builtin_nontemporal_load(partials + 2 * b + 1);
    }
    out[0] = (real)(-ts / (double)M);
    out[1] = (real)(tr / (double)M);
    *ticket = 0;
  }
}

constexpr int KE_BT = 64;

// thread = env: maha(new, old), maha(new, proj), maha(proj, old) and
// grad_mean = coeff / N * Sigma_proj^-1 (mean_new - mean_proj)
template <typename real>
__global__ __launch_bounds__(KE_BT) void kl_shared_env_kernel(
    const real* __restrict__ mn, const real* __restrict__ mo, const real* __restrict__ mp,
    const real* __restrict__ Lo, const real* __restrict__ Lp, int64_t N, int K, real gscale,
    real* __restrict__ gmean, double* __restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  real* Los = reinterpret_cast<real*>(smem_raw);     // [K][KP]
  const int KP = sm_pitch(K);
  real* Lps = Los + K * KP;
  real* vs = Lps + K * KP;                            // [KE_BT][KP]
  __shared__ double red[4];
  const int tid = threadIdx.x;
  for (int e = tid; e < K * K; e += KE_BT) {
    const int i = e / K, j = e - i * K;
    Los[i * KP + j] = Lo[e];
    Lps[i * KP + j] = Lp[e];
  }
  __syncthreads();
  const int64_t n = (int64_t)blockIdx.x * KE_BT + tid;
  const bool ok = n < N;
  const int64_t nc = ok ? n : N - 1;
  real* v = vs + tid * KP;
  // z = Lq^-1 d in place, returns |z|^2
  auto solve = [&](const real* Ls) {
    real q = 0;
    for (int r = 0; r < K; ++r) {
      real acc = v[r];
      for (int k = 0; k < r; ++k) acc -= Ls[r * KP + k] * v[k];
      acc /= Ls[r * KP + r];
      v[r] = acc;
      q += acc * acc;
    }
    return q;
  };
  for (int k = 0; k < K; ++k) v[k] = mn[nc * K + k] - mo[nc * K + k];
  double m1 = (double)solve(Los);
  for (int k = 0; k < K; ++k) v[k] = mp[nc * K + k] - mo[nc * K + k];
  double m3 = (double)solve(Los);
  for (int k = 0; k < K; ++k) v[k] = mn[nc * K + k] - mp[nc * K + k];
  double m2 = (double)solve(Lps);
  if (gmean) {                                        // w = Lp^-T z
    for (int r = K - 1; r >= 0; --r) {
      real acc = v[r];
      for (int k = r + 1; k < K; ++k) acc -= Lps[k * KP + r] * v[k];
      acc /= Lps[r * KP + r];
      v[r] = acc;
    }
    if (ok)
      for (int k = 0; k < K; ++k) gmean[n * K + k] = gscale * v[k];
  }
  if (!ok) { m1 = 0; m2 = 0; m3 = 0; }
  m1 = block_sum(m1, red);
  m2 = block_sum(m2, red);
  m3 = block_sum(m3, red);
  if (tid == 0) {
    partials[blockIdx.x * 3 + 0] = m1;
    partials[blockIdx.x * 3 + 1] = m2;
    partials[blockIdx.x * 3 + 2] = m3;
  }
}

// one workgroup: K x K parts, final sums, trust region loss and its dL
template <typename real>
__global__ __launch_bounds__(SM_BT) void kl_shared_mat_kernel(
    const real* __restrict__ Ln, const real* __restrict__ Lo, const real* __restrict__ Lp,
    int64_t N, int K, real coeff, int include_cov, const double* __restrict__ partials,
    int nparts, real* __restrict__ out, real* __restrict__ gL, int par) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int KP = sm_pitch(K);
  double* A = reinterpret_cast<double*>(smem_raw);   // Ln
  double* B = A + K * KP;                             // Lo
  double* C = B + K * KP;                             // Lp
  double* X = C + K * KP;                             // scratch
  __shared__ double red[4];
  sm_load(A, Ln, K, KP, true);
  sm_load(B, Lo, K, KP, true);
  sm_load(C, Lp, K, KP, true);
  auto frob_ld = [&](const double* Xw, double& f, double& ld) {
    double lf = 0, ll = 0;
    for (int e = threadIdx.x; e < K * K; e += SM_BT) {
      const int i = e / K, j = e - i * K;
      const double x = Xw[i * KP + j];
      lf += x * x;
      if (i == j) ll += log(x);
    }
    f = sm_block_sum(lf, red);
    ld = sm_block_sum(ll, red);
  };
  // X = Lq^-1 Ls (quad-parallel forward substitution, smallmat.h)
  auto solve = [&](double* Xw, const double* Lq, const double* Ls) {
    for (int e = threadIdx.x; e < K * K; e += SM_BT) {
      const int i = e / K, j = e - i * K;
      Xw[i * KP + j] = Ls[i * KP + j];
    }
    __syncthreads();
    sm_trsm_l(Xw, Lq, K, KP);
  };
  double f1, l1, f2, l2, f3, l3;
  (void)par;
  solve(X, B, A);                                       // new || old
  frob_ld(X, f1, l1);
  solve(X, B, C);                                       // proj || old
  frob_ld(X, f3, l3);
  solve(X, C, A);                                       // new || proj; X = Lp^-1 Ln stays
  frob_ld(X, f2, l2);
  if (gL) {
    sm_trsm_lt(X, C, K, KP);                          // Sigma_proj^-1 Ln
    for (int e = threadIdx.x; e < K * K; e += SM_BT) {
      const int i = e / K, j = e - i * K;
      double v = 0;
      if (include_cov && j <= i) v = (double)coeff * (X[i * KP + j] - (i == j ? 1.0 / A[i * KP + i] : 0.0));
      gL[e] = (real)v;
    }
  }
  double lp_ld = 0;
  for (int i = threadIdx.x; i < K; i += SM_BT) lp_ld += log(C[i * KP + i]);
  lp_ld = sm_block_sum(lp_ld, red);
  if (threadIdx.x == 0) {
    double m[3] = {0, 0, 0};
    for (int b = 0; b < nparts; ++b)
      for (int k = 0; k < 3; ++k) m[k] += partials[b * 3 + k];
    const double f[3] = {f1, f2, f3}, l[3] = {l1, l2, l3};
    double tr = 0;
    for (int k = 0; k < 3; ++k) {
      const double mean_part = 0.5 * m[k] / (double)N;
      const double shape = 0.5 * (f[k] - (double)K), volume = -l[k];
      out[4 * k + 0] = (real)mean_part;
      out[4 * k + 1] = (real)(shape + volume);
      out[4 * k + 2] = (real)shape;
      out[4 * k + 3] = (real)volume;
      if (k == 1) tr = mean_part + (include_cov ? shape + volume : 0.0);
    }
    out[12] = (real)(0.5 * (double)K * (1.0 + 1.8378770664093453) + lp_ld);   // entropy(proj)
    out[13] = (real)((double)coeff * tr);                                      // trust region loss
    out[14] = 0;
    out[15] = 0;
  }
}

template <typename real>
int surrogate(const real* lp_new, const real* lp_old, const real* adv, int64_t M, real* out,
              real* grad, double* ws, hipStream_t st) {
  TCE_CHECK_ARG(lp_new && lp_old && adv && out && ws && M > 0,
                "surrogate: null buffer / bad size");
  const unsigned grid = (unsigned)tmin<int64_t>(ceil_div(M, 4 * SUR_BT), SUR_MAX_BLOCKS);
  hipLaunchKernelGGL(surrogate_kernel<real>, dim3(grid), dim3(SUR_BT), 0, st, lp_new, lp_old, adv,
                     M, out, grad, ws + 1, reinterpret_cast<unsigned*>(ws));
  TCE_LAUNCH_CHECK();
  return 0;
}

template <typename real>
int kl_shared(const real* mn, const real* mo, const real* mp, const real* Ln, const real* Lo,
              const real* Lp, int64_t N, int K, real coeff, int include_cov, real* out,
              real* gmean, real* gL, double* ws, hipStream_t st) {
  TCE_CHECK_ARG(mn && mo && mp && Ln && Lo && Lp && out && ws && N > 0 && K > 0 && K <= 64,
                "kl_shared: bad arguments (K <= 64)");
  const int nblk = (int)ceil_div(N, KE_BT);
  const size_t lds_e = ((size_t)2 * K * sm_pitch(K) + (size_t)KE_BT * sm_pitch(K)) * sizeof(real);
  if (lds_e > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kl_shared_env_kernel<real>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_e);
  hipLaunchKernelGGL(kl_shared_env_kernel<real>, dim3(nblk), dim3(KE_BT), lds_e, st, mn, mo, mp,
                     Lo, Lp, N, K, coeff / (real)N, gmean, ws);
  TCE_LAUNCH_CHECK();
  const int par = (size_t)6 * K * sm_pitch(K) * sizeof(double) <= 150 * 1024;   // K <= 55
  const size_t lds_m = (size_t)(par ? 6 : 4) * K * sm_pitch(K) * sizeof(double);
  if (lds_m > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kl_shared_mat_kernel<real>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m);
  hipLaunchKernelGGL(kl_shared_mat_kernel<real>, dim3(1), dim3(SM_BT), lds_m, st, Ln, Lo, Lp, N,
                     K, coeff, include_cov, ws, nblk, out, gL, par);
  TCE_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" {

int64_t tce_kl_shared_ws_len(int64_t N) { return 3 * ceil_div(N, KE_BT); }

int64_t tce_surrogate_ws_len(void) { return 1 + 2 * SUR_MAX_BLOCKS; }

int tce_surrogate_f32(const float* lp_new, const float* lp_old, const float* adv, int64_t M,
                      float* out, float* grad_lp, double* ws, void* stream) {
  return surrogate<float>(lp_new, lp_old, adv, M, out, grad_lp, ws, (hipStream_t)stream);
}
int tce_surrogate_f64(const double* lp_new, const double* lp_old, const double* adv, int64_t M,
                      double* out, double* grad_lp, double* ws, void* stream) {
  return surrogate<double>(lp_new, lp_old, adv, M, out, grad_lp, ws, (hipStream_t)stream);
}
int tce_kl_shared_f32(const float* mean_new, const float* mean_old, const float* mean_proj,
                      const float* L_new, const float* L_old, const float* L_proj, int64_t N,
                      int K, float tr_coeff, int tr_include_cov, float* out16,
                      float* grad_mean, float* grad_L, double* ws, void* stream) {
  return kl_shared<float>(mean_new, mean_old, mean_proj, L_new, L_old, L_proj, N, K, tr_coeff,
                          tr_include_cov, out16, grad_mean, grad_L, ws, (hipStream_t)stream);
}
int tce_kl_shared_f64(const double* mean_new, const double* mean_old, const double* mean_proj,
                      const double* L_new, const double* L_old, const double* L_proj, int64_t N,
                      int K, double tr_coeff, int tr_include_cov, double* out16,
                      double* grad_mean, double* grad_L, double* ws, void* stream) {
  return kl_shared<double>(mean_new, mean_old, mean_proj, L_new, L_old, L_proj, N, K, tr_coeff,
                           tr_include_cov, out16, grad_mean, grad_L, ws, (hipStream_t)stream);
}

}  // extern "C"
