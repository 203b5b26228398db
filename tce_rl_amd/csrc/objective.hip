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
#include "lanevec.h"
#include "xchg.h"
#include "../../include/tce_hip.h"

namespace {

// out[0] = -mean(ratio * adv), out[1] = mean(ratio); grad[i] = -ratio_i adv_i / M.
// Grid of blocks with per-block partial sums; the block that finishes last
// adds them in a fixed order (deterministic) and re-arms the ticket.
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
  if (last) {                                         // uniform over the block
    // the partials are summed by the first wave, lane b <- blocks b, b + 64:
    // a fixed order (one thread adding 128 values was a chain of 256 dependent
    // L2 loads, 15 of the kernel's 20 us)
    __threadfence();
    double ts = 0, tr = 0;
    if (threadIdx.x < 64) {
      for (unsigned b = threadIdx.x; b < gridDim.x; b += 64) {
        ts += __builtin_nontemporal_load(partials + 2 * b);
        tr += __builtin_nontemporal_load(partials + 2 * b + 1);
      }
      ts = wave_sum(ts);
      tr = wave_sum(tr);
    }
    if (threadIdx.x == 0) {
      out[0] = (real)(-ts / (double)M);
      out[1] = (real)(tr / (double)M);
      *ticket = 0;
    }
  }
}

constexpr int KE_BT = 256;          // 4 waves
constexpr int KE_EPW = 4;           // envs per wave, one after the other
constexpr int KE_EPB = (KE_BT / 64) * KE_EPW;

// wave = env, lane = element of the mean vector (lanevec.h): maha(new, old),
// maha(new, proj), maha(proj, old) -- three forward substitutions side by side
// -- and grad_mean = coeff / N * Sigma_proj^-1 (mean_new - mean_proj)
template <typename real>
__global__ __launch_bounds__(KE_BT) void kl_shared_env_kernel(
    const real* __restrict__ mn, const real* __restrict__ mo, const real* __restrict__ mp,
    const real* __restrict__ Lo, const real* __restrict__ Lp, int64_t N, int K, real gscale,
    real* __restrict__ gmean, double* __restrict__ partials, const real* __restrict__ quad,
    real eps_mean) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  real* Los = reinterpret_cast<real*>(smem_raw);     // [K][KP]
  const int KP = sm_pitch(K);
  real* Lps = Los + K * KP;
  __shared__ double red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < K * K; e += KE_BT) {
    const int i = e / K, j = e - i * K;
    Los[i * KP + j] = Lo[e];
    Lps[i * KP + j] = Lp[e];
  }
  __syncthreads();
  const bool in = lane < K;
  const real rdo = in ? real(1) / Los[lane * KP + lane] : real(0);
  const real rdp = in ? real(1) / Lps[lane * KP + lane] : real(0);
  double m1 = 0, m2 = 0, m3 = 0;                      // this lane's share of the sums
  const int64_t n0 = (int64_t)blockIdx.x * KE_EPB + wave * KE_EPW;
  for (int i = 0; i < KE_EPW; ++i) {
    const int64_t n = n0 + i;
    if (n >= N) break;
    real a = 0, b = 0, c = 0;
    if (in) { a = mn[n * K + lane]; c = mp[n * K + lane]; }
    real v[3] = {0, 0, a - c};
    if (quad) {
      // the mean projection has formed q = |Lo^-1 (new - old)|^2 already
      // (tce_mean_proj_fwd_q_*), and proj - old = (new - old) / s: maha(new, old)
      // = q, maha(proj, old) = q / s^2 -- ONE forward substitution instead of three
      // (the kernel is bound by the instructions its waves issue, not by the
      // length of the chain: C2 policy updates/s + 3.6 %)
      const real q = quad[n];
      const real m = real(0.5) * q;
      real s2 = 1;
      if (m > eps_mean) s2 = m / eps_mean;                    // s^2
      if (lane == 0) { m1 += (double)q; m3 += (double)q / (double)s2; }
      real w1[1] = {v[2]};
      const real* const L1[1] = {Lps};
      const real r1[1] = {rdp};
      lv_solve_lower<real, 1>(w1, L1, r1, K, KP, lane);
      v[2] = w1[0];
    } else {
      if (in) b = mo[n * K + lane];
      v[0] = a - b;
      v[1] = c - b;
      const real* const Ls[3] = {Los, Los, Lps};
      const real rd[3] = {rdo, rdo, rdp};
      lv_solve_lower<real, 3>(v, Ls, rd, K, KP, lane);
      m1 += (double)v[0] * (double)v[0];
      m3 += (double)v[1] * (double)v[1];
    }
    m2 += (double)v[2] * (double)v[2];
    if (gmean) {                                      // w = Lp^-T z
      real w[1] = {v[2]};
      const real* const Lw[1] = {Lps};
      const real rw[1] = {rdp};
      lv_solve_lower_t<real, 1>(w, Lw, rw, K, KP, lane);
      if (in) gmean[n * K + lane] = gscale * w[0];
    }
  }
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
  double m[3] = {0, 0, 0};
  for (int b = threadIdx.x; b < nparts; b += SM_BT)
    for (int k = 0; k < 3; ++k) m[k] += partials[b * 3 + k];
  for (int k = 0; k < 3; ++k) m[k] = sm_block_sum(m[k], red);
  if (threadIdx.x == 0) {
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
              real* gmean, real* gL, double* ws, hipStream_t st, const real* quad = nullptr,
              real eps_mean = real(0)) {
  TCE_CHECK_ARG(mn && mo && mp && Ln && Lo && Lp && out && ws && N > 0 && K > 0 && K <= 64,
                "kl_shared: bad arguments (K <= 64)");
  const int nblk = (int)ceil_div(N, KE_EPB);
  const size_t lds_e = (size_t)2 * K * sm_pitch(K) * sizeof(real);
  if (lds_e > 48 * 1024)
    tce_lds_limit(reinterpret_cast<const void*>(kl_shared_env_kernel<real>), (size_t)(lds_e));
  hipLaunchKernelGGL(kl_shared_env_kernel<real>, dim3(nblk), dim3(KE_BT), lds_e, st, mn, mo, mp,
                     Lo, Lp, N, K, coeff / (real)N, gmean, ws, quad, eps_mean);
  TCE_LAUNCH_CHECK();
  const int par = (size_t)6 * K * sm_pitch(K) * sizeof(double) <= 150 * 1024;   // K <= 55
  const size_t lds_m = (size_t)(par ? 6 : 4) * K * sm_pitch(K) * sizeof(double);
  if (lds_m > 48 * 1024)
    tce_lds_limit(reinterpret_cast<const void*>(kl_shared_mat_kernel<real>), (size_t)(lds_m));
  hipLaunchKernelGGL(kl_shared_mat_kernel<real>, dim3(1), dim3(SM_BT), lds_m, st, Ln, Lo, Lp, N,
                     K, coeff, include_cov, ws, nblk, out, gL, par);
  TCE_LAUNCH_CHECK();
  return 0;
}


// ---------------------------------------------------------------------------
// The whole objective of one epoch in one call (tce_policy_objective_*): the
// chain of rl/objective.py `evaluate`, with the single-workgroup K x K kernels
// (covariance projection forward / backward, the matrix part of the KL
// diagnostics) on a second stream beside the per-env kernels.
// ---------------------------------------------------------------------------
struct ObjSide {
  hipStream_t side = nullptr;
  hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

int g_obj_streams = 2;            // tce_policy_objective_streams: 1 = everything on the caller's stream
int g_inline_surrogate = 1;       // tce_policy_inline_surrogate (A / B runs, tests)

// (tce_policy_objective_use_stream: a stream of the caller's choice instead of one
// created here -- a HIP stream is bound to one of the device's few hardware
// queues when it is created, and the caller can PROBE which of its streams run
// beside its other ones: tce_rl_amd/streams.py)
static hipStream_t g_obj_given = nullptr;
inline ObjSide* obj_side() {
  static ObjSide s;
  static bool tried = false, ok = false;
  if (!tried) {
    tried = true;
    ok = g_obj_given != nullptr ||
         hipStreamCreateWithFlags(&s.side, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < 7; ++i)
      ok = hipEventCreateWithFlags(&s.ev[i], hipEventDisableTiming) == hipSuccess;
  }
  if (ok && g_obj_given) s.side = g_obj_given;
  return ok ? &s : nullptr;
}

inline int64_t obj_up4(int64_t n) { return (n + 3) / 4 * 4; }
inline int64_t obj_ws_len(int64_t N, int K, int P) {
  return 3 * obj_up4(N * K) + 2 * obj_up4(N * P) + 4 * obj_up4((int64_t)K * K) + obj_up4(N);
}
template <typename real>
inline real* obj_proj_L(real* ws, int64_t N, int K, int P) {
  return ws + 3 * obj_up4(N * K) + 2 * obj_up4(N * P);
}

// a[i] += b[i] for two pairs of arrays in one launch
template <typename real>
__global__ __launch_bounds__(256) void obj_add2_kernel(real* __restrict__ a1,
                                                       const real* __restrict__ b1, int64_t n1,
                                                       real* __restrict__ a2,
                                                       const real* __restrict__ b2, int64_t n2) {
  const int64_t i = blockIdx.x * 256ll + threadIdx.x;
  if (i < n1) a1[i] += b1[i];
  else if (i - n1 < n2) a2[i - n1] += b2[i - n1];
}
// g[i,i] -= coef / L[i,i]  (gradient of -coef * entropy(L) w.r.t. L)
template <typename real>
__global__ __launch_bounds__(64) void obj_ent_diag_kernel(real* __restrict__ g,
                                                          const real* __restrict__ L, int K,
                                                          real coef) {
  const int i = threadIdx.x;
  if (i < K) g[i * K + i] -= coef / L[i * K + i];
}

// g = the gradient of -coef * entropy(L) alone: -coef / L[i,i] on the diagonal
template <typename real>
__global__ __launch_bounds__(256) void obj_ent_only_kernel(real* __restrict__ g,
                                                           const real* __restrict__ L, int K,
                                                           real coef) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= K * K) return;
  const int i = e / K, j = e - i * K;
  g[e] = i == j ? -coef / L[e] : real(0);
}
// out[0] = |g|_2 over n elements (one workgroup; fixed order)
template <typename real>
__global__ __launch_bounds__(1024) void obj_norm_kernel(const real* __restrict__ g, int64_t n,
                                                        real* __restrict__ out, real scale) {
  __shared__ double red[16];
  double sq = 0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) sq += (double)g[i] * (double)g[i];
  sq = block_sum(sq, red);
  if (threadIdx.x == 0) out[0] = (real)sqrt(sq) * scale;
}
// a[i] += b[i] (+ c[i])
template <typename real>
__global__ __launch_bounds__(256) void obj_add3_kernel(real* __restrict__ a,
                                                       const real* __restrict__ b,
                                                       const real* __restrict__ c, int64_t n) {
  const int64_t i = blockIdx.x * 256ll + threadIdx.x;
  if (i < n) a[i] += c ? b[i] + c[i] : b[i];
}

template <typename real> struct ObjApi;
template <> struct ObjApi<float> {
  static constexpr auto vec_env = tce_vec_env_f32;
  static constexpr auto mean_bwd_acc = tce_mean_proj_bwd_acc_f32;
  static constexpr auto mean_fwd_q = tce_mean_proj_fwd_q_f32;
  static constexpr auto proj_fwd = tce_kl_cov_proj_fwd_f32;
  static constexpr auto proj_bwd = tce_kl_cov_proj_bwd_f32;
  static constexpr auto pl_fwd = tce_pair_logprob_fwd_f32;
  static constexpr auto pl_bwd = tce_pair_logprob_bwd_f32;
  static constexpr auto pl_bwd_sur = tce_pair_logprob_bwd_sur_f32;
  static constexpr auto chol_fwd = tce_chol_build_fwd_f32;
  static constexpr auto logp_bwd_z_sur = tce_mvn_logprob_bwd_z_sur_f32;
};
template <> struct ObjApi<double> {
  static constexpr auto vec_env = tce_vec_env_f64;
  static constexpr auto mean_bwd_acc = tce_mean_proj_bwd_acc_f64;
  static constexpr auto mean_fwd_q = tce_mean_proj_fwd_q_f64;
  static constexpr auto proj_fwd = tce_kl_cov_proj_fwd_f64;
  static constexpr auto proj_bwd = tce_kl_cov_proj_bwd_f64;
  static constexpr auto pl_fwd = tce_pair_logprob_fwd_f64;
  static constexpr auto pl_bwd = tce_pair_logprob_bwd_f64;
  static constexpr auto pl_bwd_sur = tce_pair_logprob_bwd_sur_f64;
  static constexpr auto chol_fwd = tce_chol_build_fwd_f64;
  static constexpr auto logp_bwd_z_sur = tce_mvn_logprob_bwd_z_sur_f64;
};

#define OBJ_TRY(call)          \
  do {                         \
    const int rc_ = (call);    \
    if (rc_ != 0) return rc_;  \
  } while (0)
#define OBJ_HIP(call)                                          \
  do {                                                         \
    if (!single && (call) != hipSuccess) {                     \
      tce_set_error("policy_objective: stream / event call");  \
      return 1;                                                \
    }                                                          \
  } while (0)

#define OBJ_HIP_ALWAYS(call)                                   \
  do {                                                         \
    if ((call) != hipSuccess) {                                \
      tce_set_error("policy_objective: runtime call");         \
      return 1;                                                \
    }                                                          \
  } while (0)

template <typename real>
int policy_objective(const real* mean_new, const real* L_new, const real* mean_old,
                     const real* L_old, const real* traj, const real* logp_old,
                     const real* adv, const int64_t* pairs, const real* tab, int M, int nbg,
                     real tau, real delay, real scaled_dt, real inv_scale_g, int rel_goal,
                     const real* times, int flags_fwd, int flags_bwd, const real* t0,
                     const real* y0, const real* v0, real reg, real* basis_ws, int* flag_ws,
                     real* pl_work, real eps_mean, double eps_cov, const real* beta,
                     int entropy_eq, double* proj_ctx, real tr_coeff, int tr_include_cov,
                     real ent_coef, double* sur_ws, double* kl_ws, real* ws, real* grad_mean,
                     real* grad_L, real* sur2, real* out16, int64_t N, int T, int P, int dof,
                     int K, int proj_started, int defer_join, hipStream_t st) {
  typedef ObjApi<real> A;
  TCE_CHECK_ARG(mean_new && L_new && mean_old && L_old && traj && logp_old && adv && pairs &&
                    proj_ctx && sur_ws && kl_ws && ws && grad_mean && grad_L && sur2 && out16,
                "policy_objective: null buffer");
  TCE_CHECK_ARG(N > 0 && T > 0 && P > 0 && K == dof * nbg && K <= 64,
                "policy_objective: bad sizes (K = dof * nbg <= 64)");
  const bool single = g_obj_streams < 2;
  ObjSide* S = single ? nullptr : obj_side();
  TCE_CHECK_ARG(single || S != nullptr, "policy_objective: could not create the side stream");
  hipStream_t sd = single ? st : S->side;
  real* pm = ws;                                  // projected mean [N,K]
  real* g_pm = pm + obj_up4(N * K);               // d / d pm
  real* gm_p = g_pm + obj_up4(N * K);             // ... back through the mean projection
  real* logp = gm_p + obj_up4(N * K);             // [N,P]
  real* glp = logp + obj_up4(N * P);
  real* pL = obj_proj_L(ws, N, K, P);             // projected factor [K,K]
  real* g_pL = pL + obj_up4((int64_t)K * K);
  real* gL_p = g_pL + obj_up4((int64_t)K * K);
  // ---- fork: covariance projection (one workgroup) beside the mean projection
  // (proj_started: tce_policy_objective_begin_* has put it on the side stream)
  if (!proj_started) {
    OBJ_HIP(hipEventRecord(S->ev[0], st));
    OBJ_HIP(hipStreamWaitEvent(sd, S->ev[0], 0));
    OBJ_TRY(A::proj_fwd(L_new, L_old, 0, eps_cov, beta, entropy_eq, pL, proj_ctx, 1, K, 1, sd));
    OBJ_HIP(hipEventRecord(S->ev[1], sd));
  }
  real* quad = g_pL + 3 * obj_up4((int64_t)K * K);           // [N] |Lo^-1 (new - old)|^2
  // (gm_p: in the deferred join the projection's backward adds into grad_mean, so
  // that slot is free to carry z = Lo^-1 (new - old) from the forward to it)
  real* zbuf = ((defer_join & 1) && !(defer_join & 2)) ? gm_p : nullptr;
  OBJ_TRY(A::mean_fwd_q(mean_new, mean_old, L_old, 0, eps_mean, pm, quad, zbuf, N, K, st));
  OBJ_HIP(hipEventRecord(S->ev[2], st));
  // ---- side: KL diagnostics, entropy, trust region loss and its gradients
  OBJ_HIP(hipStreamWaitEvent(sd, S->ev[2], 0));
  OBJ_TRY(kl_shared<real>(mean_new, mean_old, pm, L_new, L_old, pL, N, K, tr_coeff,
                          tr_include_cov, out16, grad_mean, grad_L, kl_ws, sd, quad, eps_mean));
  // grad_mean (written by kl_shared_env_kernel on the side stream) is complete
  // here: the deferred join adds to it on `st` before ev[4] is waited for
  OBJ_HIP(hipEventRecord(S->ev[5], sd));
  // ---- main: pair log-prob under the projection, surrogate, and back
  OBJ_HIP(hipStreamWaitEvent(st, S->ev[1], 0));
  const bool split = (defer_join & 2) != 0;
  // The backward pair kernels recompute the log-prob and can form the surrogate's
  // gradient themselves and leave the log-probs behind
  // (tce_pair_logprob_bwd_sur_*): no forward pass at all (C2: 32 us of all-CU
  // time per epoch, 83 at K 63), and the surrogate sums, which then only feed the
  // record row, run on the side stream.  Where the shared-factor
  // fast path is known to run (what the deferred join's caller, the direct
  // epoch, guarantees with flag bit 3); with one stream the same kernels in
  // the same roles, one after the other.
  const bool inline_sur = g_inline_surrogate && (defer_join & 1) && !split &&
                          (flags_fwd & 1) == 0 && (flags_fwd & 8) != 0 && (flags_bwd & 8) != 0 &&
                          N >= 256;
  if (inline_sur) {
    // (the backward call runs pair_prep: the forward's flags said so)
    OBJ_TRY(A::pl_bwd_sur(traj, pm, pL, 0, pairs, tab, M, nbg, tau, delay, scaled_dt, inv_scale_g,
                          rel_goal, times, flags_fwd, t0, y0, v0, reg, logp_old, adv, logp, g_pm,
                          g_pL, basis_ws, flag_ws, pl_work, N, T, P, dof, st));
  } else {
    OBJ_TRY(A::pl_fwd(traj, pm, pL, 0, pairs, tab, M, nbg, tau, delay, scaled_dt, inv_scale_g,
                      rel_goal, times, flags_fwd, t0, y0, v0, reg, logp, basis_ws, flag_ws,
                      pl_work, N, T, P, dof, st));
    OBJ_TRY(surrogate<real>(logp, logp_old, adv, N * (int64_t)P, sur2, glp, sur_ws, st));
    OBJ_TRY(A::pl_bwd(traj, pm, pL, 0, pairs, tab, M, nbg, tau, delay, scaled_dt, inv_scale_g,
                      rel_goal, times, flags_bwd, t0, y0, v0, reg, glp, g_pm, g_pL, basis_ws,
                      flag_ws, pl_work, N, T, P, dof, st));
  }
  if (ent_coef != real(0) && !split) {
    hipLaunchKernelGGL(obj_ent_diag_kernel<real>, dim3(1), dim3(64), 0, st, g_pL, pL, K,
                       ent_coef);
    TCE_LAUNCH_CHECK();
  }
  OBJ_HIP(hipEventRecord(S->ev[3], st));
  // ---- back through the two projections, side by side
  OBJ_HIP(hipStreamWaitEvent(sd, S->ev[3], 0));
  OBJ_TRY(A::proj_bwd(L_new, L_old, 0, pL, proj_ctx, g_pL, gL_p, 1, K, sd));
  if (split && ent_coef != real(0)) {
    // the entropy term's own way back (gL_p above is the surrogate's alone)
    hipLaunchKernelGGL(obj_ent_only_kernel<real>, dim3((unsigned)ceil_div((int64_t)K * K, 256)),
                       dim3(256), 0, sd, g_pL, pL, K, ent_coef);
    TCE_LAUNCH_CHECK();
    OBJ_TRY(A::proj_bwd(L_new, L_old, 0, pL, proj_ctx, g_pL, gL_p + obj_up4((int64_t)K * K), 1,
                        K, sd));
  }
  if (inline_sur) {
    // the loss value for the record: the surrogate sums over the log-probs the
    // backward kernels left behind, on the side stream behind the projection's
    // backward (which waited for ev[3], i.e. for those kernels)
    OBJ_TRY(surrogate<real>(logp, logp_old, adv, N * (int64_t)P, sur2, (real*)nullptr, sur_ws, sd));
  }
  OBJ_HIP(hipEventRecord(S->ev[4], sd));
  const bool split0 = (defer_join & 2) != 0;
  if ((defer_join & 1) && !split0) {
    // deferred join: the mean projection's backward adds to grad_mean itself
    // (what obj_add2_kernel did in a launch of its own, same operands and order);
    // grad_L lacks the projection's part until tce_policy_objective_end_*
    OBJ_HIP(hipStreamWaitEvent(st, S->ev[5], 0));
    return A::mean_bwd_acc(mean_new, mean_old, L_old, 0, eps_mean, g_pm, zbuf, grad_mean, N, K,
                           st);
  }
  OBJ_TRY(A::vec_env(1, 1, mean_new, mean_old, L_old, 0, eps_mean, g_pm, nullptr, gm_p,
                     nullptr, N, K, st));
  const int64_t n1 = N * K, n2 = (int64_t)K * K;
  if (split) {
    // nothing is added: grad_mean / grad_L hold the trust region loss's gradient,
    // ws the surrogate's (mean: gm_p; factor: gL_p once ev[4] has passed)
    OBJ_HIP(hipStreamWaitEvent(st, S->ev[5], 0));
    return 0;
  }
  if (defer_join & 1) {
    // grad_mean is complete after this; grad_L lacks the projection's part until
    // tce_policy_objective_end_* joins the side stream
    OBJ_HIP(hipStreamWaitEvent(st, S->ev[5], 0));
    hipLaunchKernelGGL(obj_add2_kernel<real>, dim3((unsigned)ceil_div(n1, 256)), dim3(256), 0, st,
                       grad_mean, gm_p, n1, grad_L, gL_p, (int64_t)0);
    TCE_LAUNCH_CHECK();
    return 0;
  }
  // ---- join
  OBJ_HIP(hipStreamWaitEvent(st, S->ev[4], 0));
  hipLaunchKernelGGL(obj_add2_kernel<real>, dim3((unsigned)ceil_div(n1 + n2, 256)), dim3(256), 0,
                     st, grad_mean, gm_p, n1, grad_L, gL_p, n2);
  TCE_LAUNCH_CHECK();
  return 0;
}

template <typename real>
int policy_objective_end(real* grad_L, real* ws, int64_t N, int K, int P, hipStream_t st) {
  TCE_CHECK_ARG(grad_L && ws && N > 0 && K > 0 && K <= 64, "policy_objective_end: bad arguments");
  const bool single = g_obj_streams < 2;
  ObjSide* S = single ? nullptr : obj_side();
  TCE_CHECK_ARG(single || S != nullptr, "policy_objective: could not create the side stream");
  OBJ_HIP(hipStreamWaitEvent(st, S->ev[4], 0));
  const int64_t n2 = (int64_t)K * K;
  real* gL_p = obj_proj_L(ws, N, K, P) + 2 * obj_up4(n2);
  hipLaunchKernelGGL(obj_add2_kernel<real>, dim3((unsigned)ceil_div(n2, 256)), dim3(256), 0, st,
                     grad_L, gL_p, n2, grad_L, gL_p, (int64_t)0);
  TCE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------
// The same objective for the black-box agent (black_box_agent.py:283-357): the
// log-prob is the one of the sampled parameter vectors under the projected
// Gaussian (tce_vec_env mode 2) instead of the pair-wise trajectory log-prob.
// ---------------------------------------------------------------------------

// pm | g_pm | gm_p [N,K] | logp | glp [N] | pL | g_pL | gL_p | gL_e [K,K] | z [N,K] |
// slabs of the factor's gradient [ceil(N / 32)][K K + K] | column sums [K] |
// |L_old^-1 (new - old)|^2 [N] | L_old^-1 (new - old) [N,K]
constexpr int BB_OL_ROWS = 32;                    // = OL_ROWS (out_layer_grad's rows per slab)
inline int64_t bb_obj_ws_len(int64_t N, int K) {
  const int64_t KK = (int64_t)K * K;
  return 3 * obj_up4(N * K) + 2 * obj_up4(N) + 4 * obj_up4(KK) + obj_up4(N * K) +
         obj_up4(ceil_div(N, BB_OL_ROWS) * (KK + K)) + obj_up4(K) + obj_up4(N) + obj_up4(N * K);
}

template <typename real>
int out_layer_grad(const real* g, const real* h, real* dW, real* db, real* ws, int64_t N, int K,
                   int H, hipStream_t st);

// g_pL <- tril(g_pL) - (sum_g[0] + ent) diag(1 / L_ii): the shared factor's gradient
// from the product (g q)^T z (tce_mvn_logprob_bwd_z_*), sum_g[0] = sum_n d loss / d
// logp_n (= the surrogate loss itself), ent = the entropy bonus' coefficient
template <typename real>
__global__ __launch_bounds__(256) void bb_gpl_fix_kernel(real* __restrict__ g,
                                                         const real* __restrict__ L,
                                                         const real* __restrict__ sum_g, int K,
                                                         real ent) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= K * K) return;
  const int i = e / K, j = e - i * K;
  real v = j <= i ? g[e] : real(0);
  if (i == j) v -= (sum_g[0] + ent) / L[e];
  g[e] = v;
}

template <typename real>
int bb_policy_objective(const real* mean_new, const real* L_new, const real* mean_old,
                        const real* L_old, const real* actions, const real* logp_old,
                        const real* adv, real eps_mean, double eps_cov, const real* beta,
                        int entropy_eq, double* proj_ctx, real tr_coeff, int tr_include_cov,
                        real ent_coef, double* sur_ws, double* kl_ws, real* ws, real* grad_mean,
                        real* grad_L, real* sur2, real* out16, real* proj_mean_out,
                        real* proj_L_out, int64_t N, int K, hipStream_t st, int proj_started = 0,
                        int split = 0) {
  typedef ObjApi<real> A;
  TCE_CHECK_ARG(mean_new && L_new && mean_old && L_old && actions && logp_old && adv &&
                    proj_ctx && sur_ws && kl_ws && ws && grad_mean && grad_L && sur2 && out16,
                "bb_policy_objective: null buffer");
  TCE_CHECK_ARG(N > 0 && K > 0 && K <= 64, "bb_policy_objective: bad sizes (K <= 64)");
  const bool single = g_obj_streams < 2;
  ObjSide* S = single ? nullptr : obj_side();
  TCE_CHECK_ARG(single || S != nullptr, "policy_objective: could not create the side stream");
  hipStream_t sd = single ? st : S->side;
  const int64_t KK = (int64_t)K * K;
  real* pm = ws;
  real* g_pm = pm + obj_up4(N * K);
  real* gm_p = g_pm + obj_up4(N * K);
  real* logp = gm_p + obj_up4(N * K);
  real* glp = logp + obj_up4(N);
  real* pL = glp + obj_up4(N);
  real* g_pL = pL + obj_up4(KK);
  real* gL_p = g_pL + obj_up4(KK);
  real* gL_e = gL_p + obj_up4(KK);                // split: the entropy term's d / d L_new
  real* zbuf = gL_e + obj_up4(KK);                // [N,K] L_proj^-1 (actions - proj mean)
  real* ol_ws = zbuf + obj_up4(N * K);            // slabs of (g q)^T z
  real* dbs = ol_ws + obj_up4(ceil_div(N, BB_OL_ROWS) * (KK + K));
  real* quad = dbs + obj_up4(K);                  // [N] |L_old^-1 (new - old)|^2
  real* zmean = quad + obj_up4(N);                // [N,K] L_old^-1 (new - old)
  // (proj_started: bb_policy_epoch has put the Cholesky head and the covariance
  // projection on the side stream, beside the mean net's forward)
  if (!proj_started) {
    OBJ_HIP(hipEventRecord(S->ev[0], st));
    OBJ_HIP(hipStreamWaitEvent(sd, S->ev[0], 0));
    OBJ_TRY(A::proj_fwd(L_new, L_old, 0, eps_cov, beta, entropy_eq, pL, proj_ctx, 1, K, 1, sd));
    OBJ_HIP(hipEventRecord(S->ev[1], sd));
  }
  // (as the TCE objective: the mean projection leaves q = |L_old^-1 (new - old)|^2
  // and z = L_old^-1 (new - old) per env -- the KL diagnostics take maha(new, old)
  // = q and maha(proj, old) = q / s^2 from it, one substitution per env instead of
  // three, and the projection's backward starts from z)
  OBJ_TRY(A::mean_fwd_q(mean_new, mean_old, L_old, 0, eps_mean, pm, quad, zmean, N, K, st));
  OBJ_HIP(hipEventRecord(S->ev[2], st));
  OBJ_HIP(hipStreamWaitEvent(sd, S->ev[2], 0));
  OBJ_TRY(kl_shared<real>(mean_new, mean_old, pm, L_new, L_old, pL, N, K, tr_coeff,
                          tr_include_cov, out16, grad_mean, grad_L, kl_ws, sd, quad, eps_mean));
  // grad_mean (the trust region loss's part, written on the side stream) is
  // complete here: the mean projection's backward adds to it
  OBJ_HIP(hipEventRecord(S->ev[5], sd));
  OBJ_HIP(hipStreamWaitEvent(st, S->ev[1], 0));
  // log N(actions; pm, pL pL^T), surrogate, and back: ONE per-env kernel (the
  // backward recomputes z anyway and the surrogate's gradient needs no sum over
  // the envs), then the loss value from the log-probs it left behind
  // d / d proj mean per env and d / d L_proj summed over the envs: the shared factor
  // makes the sum ONE [K x N] . [N x K] product (g q)^T z (as the mean net's output
  // layer: 32-row slabs, fixed-order reduction) instead of N outer products
  // written to and re-read from HBM ([N,K,K]: 65 MB at 4096 envs and K 63, whose
  // sum alone took 209 us of a 0.8 ms epoch)
  OBJ_TRY(A::logp_bwd_z_sur(actions, pm, pL, logp_old, adv, g_pm, zbuf, logp, N, K, st));
  OBJ_TRY(surrogate<real>(logp, logp_old, adv, N, sur2, (real*)nullptr, sur_ws, st));
  OBJ_TRY(out_layer_grad<real>(g_pm, zbuf, g_pL, dbs, ol_ws, N, K, K, st));
  hipLaunchKernelGGL(bb_gpl_fix_kernel<real>, dim3((unsigned)ceil_div(KK, 256)), dim3(256), 0, st,
                     g_pL, pL, (const real*)sur2, K, split ? real(0) : ent_coef);
  TCE_LAUNCH_CHECK();
  OBJ_HIP(hipEventRecord(S->ev[3], st));
  OBJ_HIP(hipStreamWaitEvent(sd, S->ev[3], 0));
  OBJ_TRY(A::proj_bwd(L_new, L_old, 0, pL, proj_ctx, g_pL, gL_p, 1, K, sd));
  if (split && ent_coef != real(0)) {
    // the entropy term's own way back (gL_p above is the surrogate's alone)
    hipLaunchKernelGGL(obj_ent_only_kernel<real>, dim3((unsigned)ceil_div(KK, 256)), dim3(256), 0,
                       sd, g_pL, pL, K, ent_coef);
    TCE_LAUNCH_CHECK();
    OBJ_TRY(A::proj_bwd(L_new, L_old, 0, pL, proj_ctx, g_pL, gL_e, 1, K, sd));
  }
  OBJ_HIP(hipEventRecord(S->ev[4], sd));
  if (!split) {
    // the surrogate's half of d / d mean_new added to the trust region's by the
    // mean projection's backward itself, then the two halves of d / d L_new
    OBJ_HIP(hipStreamWaitEvent(st, S->ev[5], 0));
    OBJ_TRY(A::mean_bwd_acc(mean_new, mean_old, L_old, 0, eps_mean, g_pm, zmean, grad_mean, N, K,
                            st));
    OBJ_HIP(hipStreamWaitEvent(st, S->ev[4], 0));
    hipLaunchKernelGGL(obj_add2_kernel<real>, dim3((unsigned)ceil_div(KK, 256)), dim3(256), 0, st,
                       grad_L, gL_p, KK, grad_L, gL_p, (int64_t)0);
    TCE_LAUNCH_CHECK();
    if (proj_mean_out)
      OBJ_HIP_ALWAYS(hipMemcpyAsync(proj_mean_out, pm, sizeof(real) * N * K,
                                    hipMemcpyDeviceToDevice, st));
    if (proj_L_out)
      OBJ_HIP_ALWAYS(hipMemcpyAsync(proj_L_out, pL, sizeof(real) * KK, hipMemcpyDeviceToDevice,
                                    st));
    return 0;
  }
  // split: nothing is added -- grad_mean / grad_L hold the trust region loss's
  // gradient, ws the surrogate's (gm_p, gL_p) and the entropy term's (gL_e [K,K])
  OBJ_TRY(A::vec_env(1, 1, mean_new, mean_old, L_old, 0, eps_mean, g_pm, nullptr, gm_p,
                     nullptr, N, K, st));
  OBJ_HIP(hipStreamWaitEvent(st, S->ev[4], 0));
  if (proj_mean_out)
    OBJ_HIP_ALWAYS(hipMemcpyAsync(proj_mean_out, pm, sizeof(real) * N * K,
                                  hipMemcpyDeviceToDevice, st));
  if (proj_L_out)
    OBJ_HIP_ALWAYS(hipMemcpyAsync(proj_L_out, pL, sizeof(real) * KK, hipMemcpyDeviceToDevice, st));
  return 0;
}

// Cholesky head + covariance projection of the coming objective call on the
// side stream: they depend on the variance parameters only, so they can run
// beside the forward pass of the mean net.
template <typename real>
int policy_objective_begin(const real* var_vec, int nvec, real min_std, const real* L_old,
                           double eps_cov, const real* beta, int entropy_eq, double* proj_ctx,
                           real* L_new, real* ws, int64_t N, int K, int P, hipStream_t st) {
  typedef ObjApi<real> A;
  TCE_CHECK_ARG(var_vec && L_old && proj_ctx && L_new && ws && N > 0 && K > 0 && K <= 64,
                "policy_objective_begin: bad arguments");
  const bool single = g_obj_streams < 2;
  ObjSide* S = single ? nullptr : obj_side();
  TCE_CHECK_ARG(single || S != nullptr, "policy_objective: could not create the side stream");
  hipStream_t sd = single ? st : S->side;
  OBJ_HIP(hipEventRecord(S->ev[0], st));
  OBJ_HIP(hipStreamWaitEvent(sd, S->ev[0], 0));
  OBJ_TRY(A::chol_fwd(var_vec, L_new, 1, K, nvec, min_std, sd));
  OBJ_TRY(A::proj_fwd(L_new, L_old, 0, eps_cov, beta, entropy_eq, obj_proj_L(ws, N, K, P),
                      proj_ctx, 1, K, 1, sd));
  OBJ_HIP(hipEventRecord(S->ev[1], sd));
  return 0;
}

// ---------------------------------------------------------------------------
// Output layer of the mean net, y = h W^T + b, from g = dL/dy [N,K] and the
// hidden activations h [N,H]: dW [K,H] = g^T h, db [K] = sum_n g.  (As a
// library GEMM this [K x N] . [N x H] product with K = 24 ran on 4 workgroups:
// 26 us.)  Blocks of OL_ROWS rows -> partial [K H + K] slabs -> one reduction.
// ---------------------------------------------------------------------------
constexpr int OL_ROWS = 32, OL_MAXK = 64;
static_assert(OL_ROWS == BB_OL_ROWS, "bb_obj_ws_len sizes the slabs of out_layer_grad");

template <typename real>
__global__ __launch_bounds__(256) void out_layer_grad_kernel(const real* __restrict__ g,
                                                             const real* __restrict__ h,
                                                             int64_t N, int K, int H,
                                                             real* __restrict__ part) {
  __shared__ real gs[OL_ROWS][OL_MAXK + 1];
  const int j = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * OL_ROWS;
  const int nr = (int)tmin<int64_t>(OL_ROWS, N - r0);
  const int K8 = (K + 7) & ~7;
  for (int e = threadIdx.x; e < OL_ROWS * K8; e += blockDim.x) {
    const int r = e / K8, k = e - r * K8;
    gs[r][k] = (r < nr && k < K) ? g[(r0 + r) * K + k] : real(0);
  }
  __syncthreads();
  real* out = part + (int64_t)blockIdx.x * ((int64_t)K * H + K);
  if (j < H) {
    real hb[OL_ROWS];                                  // all loads of the block in flight
#pragma unroll
    for (int r = 0; r < OL_ROWS; ++r) hb[r] = r < nr ? h[(r0 + r) * H + j] : real(0);
    // 8 outputs at a time (columns of gs past K are zero)
    for (int k0 = 0; k0 < K; k0 += 8) {
      real acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int r = 0; r < OL_ROWS; ++r)
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) acc[kk] += gs[r][k0 + kk] * hb[r];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk)
        if (k0 + kk < K) out[(int64_t)(k0 + kk) * H + j] = acc[kk];
    }
  }
  if (j < K) {
    real sb = 0;
    for (int r = 0; r < OL_ROWS; ++r) sb += gs[r][j];
    out[(int64_t)K * H + j] = sb;
  }
}

// dst[e] = sum_b part[b][e] (fixed order, 32 loads in flight -- with 8 the 128
// slabs of a C2 epoch were 16 dependent round trips, 28 us beside the critic);
// e < KH -> dW, else db
template <typename real>
__global__ __launch_bounds__(256) void out_layer_reduce_kernel(const real* __restrict__ part,
                                                               int nb, int64_t KH, int K,
                                                               real* __restrict__ dW,
                                                               real* __restrict__ db) {
  const int64_t e = blockIdx.x * 256ll + threadIdx.x;
  const int64_t M = KH + K;
  if (e >= M) return;
  constexpr int U = 32;
  real a[U];
#pragma unroll
  for (int u = 0; u < U; ++u) a[u] = 0;
  int b = 0;
  for (; b + U <= nb; b += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] += part[(int64_t)(b + u) * M + e];
  }
  for (; b < nb; ++b) a[0] += part[(int64_t)b * M + e];
#pragma unroll
  for (int w = U / 2; w >= 1; w >>= 1)
#pragma unroll
    for (int u = 0; u < w; ++u) a[u] += a[u + w];
  if (e < KH) dW[e] = a[0];
  else db[e - KH] = a[0];
}

template <typename real>
int out_layer_grad(const real* g, const real* h, real* dW, real* db, real* ws, int64_t N, int K,
                   int H, hipStream_t st) {
  TCE_CHECK_ARG(g && h && dW && db && ws && N > 0 && K > 0 && K <= OL_MAXK && H > 0 && H <= 256,
                "out_layer_grad: bad arguments (K <= 64, H <= 256)");
  const int nb = (int)ceil_div(N, OL_ROWS);
  hipLaunchKernelGGL(out_layer_grad_kernel<real>, dim3(nb), dim3(H <= 128 ? 128 : 256), 0, st, g,
                     h, N, K, H, ws);
  TCE_LAUNCH_CHECK();
  const int64_t KH = (int64_t)K * H;
  hipLaunchKernelGGL(out_layer_reduce_kernel<real>, dim3((unsigned)ceil_div(KH + K, 256)),
                     dim3(256), 0, st, ws, nb, KH, K, dW, db);
  TCE_LAUNCH_CHECK();
  return 0;
}

// record row of one policy epoch: {surrogate, entropy loss, trust region loss,
// total, entropy, |g|, |g| clipped, 12 KL means}
template <typename real>
__global__ __launch_bounds__(64) void policy_record_kernel(const real* __restrict__ sur2,
                                                           const real* __restrict__ out16,
                                                           const real* __restrict__ norms2,
                                                           real ent_coef,
                                                           real* __restrict__ row19) {
  const int i = threadIdx.x;
  if (i >= 19) return;
  const real entl = ent_coef == real(0) ? real(0) : -ent_coef * out16[12];
  real v;
  if (i == 0) v = sur2[0];
  else if (i == 1) v = entl;
  else if (i == 2) v = out16[13];
  else if (i == 3) v = ent_coef == real(0) ? sur2[0] + out16[13] : sur2[0] + out16[13] + entl;
  else if (i == 4) v = out16[12];
  else if (i < 7) v = norms2[i - 5];
  else v = out16[i - 7];
  row19[i] = v;
}


// ---------------------------------------------------------------------------
// One whole TCE policy epoch without autograd in ONE call (what
// rl/objective.py:DirectEpoch.run issued as 12 separate calls: the epoch was
// host-bound at 25 - 35 us per call): Cholesky head + covariance projection on
// the second stream, mean net forward, the objective and its gradient (deferred
// join), mean net backward into the flat gradient, join, Cholesky head
// backward, flat Adam (do_adam; a sharded caller all-reduces first), record.
// net_kind 0: D_in <= 40 -> 128 -> 128 -> K float32 on the fused MFMA kernels
// of csrc/mlp.hip + the row kernel of the output layer; 1: any shape of
// csrc/pmlp.hip (float32 / float64, one or two hidden layers).
// balance != 0: the epoch of a balance-check iteration
// (mprl/rl/agent/temporal_correlated_agent.py:447-522): the objective is
// evaluated ONCE with its gradient split into the surrogate's and the trust
// region loss's parts (the reference runs three forward / backward passes); each
// part goes back through the mean net and the Cholesky head by itself and its
// parameter-gradient norm is stored (bal2), their sum (+ the entropy term's) is
// the epoch's gradient.
// ---------------------------------------------------------------------------
// The tail of a policy epoch in ONE launch (it was five: the join's add of the
// two halves of d / d L, the Cholesky head's backward, the two kernels of the
// flat Adam step, the record row -- 42 us of kernels and five launch boundaries
// of a 470 us epoch at C2, every one a dependent launch of a few workgroups).
// As tce_adam_once_*: every workgroup forms the head's gradient (<= K (K + 1) / 2
// values, LDS) and |g|^2 over the WHOLE flat gradient in the order of
// adam_prep_kernel (so the step is bit-identical to the five-launch tail), then
// applies its own slice.  The step count stays on the device (graph replays):
// every workgroup reads state[0] at its start, the LAST one to finish (ticket)
// writes the state vector and the record row -- a workgroup that starts late
// must not see the new count.  gL_p == nullptr: g_L is complete (balance epochs).
constexpr int PT_BT = 1024, PT_MAX_BLOCKS = 32;
// Who applies what: the mean net's part [0, PN) is split over the workgroups.
// The head's part [PN, n) -- whose gradient every workgroup forms from the
// variance PARAMETERS (sigmoid of var) for the norm -- is applied by the LAST
// workgroup to finish, i.e. after every workgroup has read var: a workgroup
// that starts late (the grid is spread over 8 XCDs beside the critic's
// persistent grid) never sees parameters this launch has already written.
// Sharded run (xchg_on(X); clip == 0, the caller's condition): every workgroup
// publishes its slice of the local gradient, waits for the peers' same
// workgroup and adds their values in rank order before Adam (csrc/xchg.h);
// only workgroup 0 reads var and owns the head's part; |g| of the summed
// gradient comes from per-workgroup partial sums added in workgroup order by the
// last one.
template <typename real>
__global__ __launch_bounds__(PT_BT) void policy_tail_kernel(
    real* __restrict__ param, real* __restrict__ grad, real* __restrict__ m, real* __restrict__ v,
    int64_t PN, int nvec, int K, const real* __restrict__ g_L, const real* __restrict__ gL_p,
    real* __restrict__ state, unsigned* __restrict__ ticket, const real* __restrict__ sur2,
    const real* __restrict__ out16, real ent_coef, real* __restrict__ row19, real lr, real b1,
    real b2, real eps, real wd, real clip, real gscale, XchgView X, double* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) char pt_smem[];
  real* gv = reinterpret_cast<real*>(pt_smem);               // [nvec] the head's gradient
  __shared__ real red[16];
  __shared__ int last_s;
  const int tid = threadIdx.x;
  const real* var = param + PN;
  const real step = state[0] + real(1);
  const bool shard = xchg_on(X);
  // ---- Cholesky head backward (chol_build_bwd_kernel) of g_L (+ gL_p)
  if (!shard || blockIdx.x == 0) {
    for (int i = tid; i < nvec; i += PT_BT) {
      int idx;
      real sig = 1;
      if (i < K) {
        idx = i * K + i;
        const real x = var[i];
        sig = x > real(20) ? real(1) : real(1) / (real(1) + exp(-x));
      } else {
        const int t = i - K;
        int r = (int)((1.0 + sqrt(1.0 + 8.0 * (double)t)) * 0.5);
        while (r * (r - 1) / 2 > t) --r;
        while ((r + 1) * r / 2 <= t) ++r;
        idx = r * K + (t - r * (r - 1) / 2);
      }
      real g = g_L[idx];
      if (gL_p) g += gL_p[idx];
      gv[i] = i < K ? g * sig : g;
    }
  }
  __syncthreads();
  const int64_t n = PN + nvec;
  real step_size, bc2s;
  adam_coef(lr, b1, b2, step, step_size, bc2s);
  const int64_t per = (PN + gridDim.x - 1) / gridDim.x;
  const int64_t i0 = blockIdx.x * per, i1 = tmin<int64_t>(PN, i0 + per);
  real before, coef = 1, cg = gscale;
  if (!shard) {
    // ---- |g|^2 over the flat gradient (adam_prep_kernel's order)
    real sq = 0;
    for (int64_t i = tid; i < n; i += PT_BT) {
      const real g = i < PN ? grad[i] : gv[i - PN];
      sq += g * g;
    }
    sq = block_sum(sq, red);
    before = sqrt(sq) * gscale;
    if (clip > real(0)) coef = tmin(clip / (before + real(1e-6)), real(1));
    cg = coef * gscale;
    // ---- this workgroup's slice of the Adam step
    for (int64_t i = i0 + tid; i < i1; i += PT_BT) {
      real w = param[i], mi = m[i], vi = v[i];
      adam_elem(grad[i] * cg, w, mi, vi, b1, b2, eps, wd, step_size, bc2s);
      m[i] = mi;
      v[i] = vi;
      param[i] = w;
    }
  } else {
    // ---- publish the local gradient, meet the peers' same workgroup, add
    for (int64_t i = i0 + tid; i < i1; i += PT_BT) xchg_put<real>(X, i, grad[i]);
    if (blockIdx.x == 0)
      for (int i = tid; i < nvec; i += PT_BT) xchg_put<real>(X, PN + i, gv[i]);
    xchg_sync(X, blockIdx.x);
    real sq = 0;
    for (int64_t i = i0 + tid; i < i1; i += PT_BT) {
      const real g = xchg_get<real>(X, i, grad[i]);
      grad[i] = g;
      sq += g * g;
      real w = param[i], mi = m[i], vi = v[i];
      adam_elem(g * cg, w, mi, vi, b1, b2, eps, wd, step_size, bc2s);
      m[i] = mi;
      v[i] = vi;
      param[i] = w;
    }
    if (blockIdx.x == 0)
      for (int i = tid; i < nvec; i += PT_BT) {
        const int64_t e = PN + i;
        const real g = xchg_get<real>(X, e, gv[i]);
        grad[e] = g;
        sq += g * g;
        real w = param[e], mi = m[e], vi = v[e];
        adam_elem(g * cg, w, mi, vi, b1, b2, eps, wd, step_size, bc2s);
        m[e] = mi;
        v[e] = vi;
        param[e] = w;
      }
    sq = block_sum(sq, red);
    if (tid == 0) partial[blockIdx.x] = (double)sq;
    before = 0;
  }
  // ---- the last workgroup to get here: the head's part, state vector, record row
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    const unsigned t = atomicAdd(ticket, 1u);
    last_s = t == gridDim.x - 1;
  }
  __syncthreads();
  if (!last_s) return;
  __threadfence();
  if (!shard) {
    for (int i = tid; i < nvec; i += PT_BT) {
      const int64_t e = PN + i;
      const real g0 = gv[i];
      grad[e] = g0;                                            // (the flat buffer keeps the head's part)
      real w = param[e], mi = m[e], vi = v[e];
      adam_elem(g0 * cg, w, mi, vi, b1, b2, eps, wd, step_size, bc2s);
      m[e] = mi;
      v[e] = vi;
      param[e] = w;
    }
  } else {
    double tot = 0;
    for (unsigned b = 0; b < gridDim.x; ++b)
      tot += __hip_atomic_load(partial + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    before = (real)sqrt(tot) * gscale;
  }
  if (tid == 0) {
    state[0] = step;
    state[1] = before;
    state[2] = before * coef;
    state[3] = cg;
    *ticket = 0u;
  }
  if (tid < 19) {
    const int i = tid;
    const real entl = ent_coef == real(0) ? real(0) : -ent_coef * out16[12];
    real r;
    if (i == 0) r = sur2[0];
    else if (i == 1) r = entl;
    else if (i == 2) r = out16[13];
    else if (i == 3) r = ent_coef == real(0) ? sur2[0] + out16[13] : sur2[0] + out16[13] + entl;
    else if (i == 4) r = out16[12];
    else if (i == 5) r = before;
    else if (i == 6) r = before * coef;
    else r = out16[i - 7];
    row19[i] = r;
  }
}
// (A / B runs and tests: the five-launch tail)
static int g_policy_tail_fused = 1;

template <typename real> struct EpApi;
template <> struct EpApi<float> {
  static constexpr auto begin = tce_policy_objective_begin_f32;
  static constexpr auto objective = tce_policy_objective_f32;
  static constexpr auto chol_bwd = tce_chol_build_bwd_f32;
  static constexpr auto adam = tce_adam_flat_f32;
  static constexpr auto record = tce_policy_record_f32;
  static constexpr auto pm_fwd = tce_pmlp_forward_f32;
  static constexpr auto pm_bwd = tce_pmlp_backward_f32;
  static constexpr auto xsum = tce_xchg_allreduce_f32;
  static constexpr auto xsum_to = tce_xchg_allreduce_to_f32;
};
template <> struct EpApi<double> {
  static constexpr auto begin = tce_policy_objective_begin_f64;
  static constexpr auto objective = tce_policy_objective_f64;
  static constexpr auto chol_bwd = tce_chol_build_bwd_f64;
  static constexpr auto adam = tce_adam_flat_f64;
  static constexpr auto record = tce_policy_record_f64;
  static constexpr auto pm_fwd = tce_pmlp_forward_f64;
  static constexpr auto pm_bwd = tce_pmlp_backward_f64;
  static constexpr auto xsum = tce_xchg_allreduce_f64;
  static constexpr auto xsum_to = tce_xchg_allreduce_to_f64;
};

inline int64_t epoch2_ws_len(int64_t N, int K, int H, int64_t nparam) {
  return 2 * obj_up4(N * (int64_t)H) + 2 * obj_up4(N * (int64_t)K) +
         3 * obj_up4((int64_t)K * K) + 36 + 2 * obj_up4(nparam);
}

// the fused 128 x 2 float32 kernels (net_kind 0) exist in float32 only
inline int fast_fwd(const float* x, int64_t xs, int64_t N, int din, int act, const float* param,
                    float* h2, float* mean, int K, void* st) {
  const float* w1 = param;
  const float* b1 = w1 + (int64_t)128 * din;
  const float* w2 = b1 + 128;
  const float* b2 = w2 + 128 * 128;
  const float* w3 = b2 + 128;
  const float* b3 = w3 + (int64_t)K * 128;
  OBJ_TRY(tce_mlp_hidden_f32(x, 0, xs, (int)N, N, din, w1, b1, w2, b2, act, nullptr, h2, nullptr,
                             nullptr, nullptr, st));
  return tce_lin_rows_f32(h2, 128, N, 128, K, w3, 1, b3, mean, st);
}
inline int fast_fwd(const double*, int64_t, int64_t, int, int, const double*, double*, double*, int,
                    void*) {
  tce_set_error("policy_epoch: net_kind 0 is float32 only");
  return 1;
}
inline int fast_bwd(const float* x, int64_t xs, int64_t N, int din, int act, const float* param,
                    const float* h2, const float* g_mean, float* gh, float* partials, float* ol_ws,
                    float* stats, float* grad, int K, void* st) {
  const float* w1 = param;
  const float* b1 = w1 + (int64_t)128 * din;
  const float* w2 = b1 + 128;
  const float* b2 = w2 + 128 * 128;
  const float* w3 = b2 + 128;
  const int64_t PH = (int64_t)128 * din + 128 + 128 * 128 + 128;
  // (the hidden layers' launch fills [0, PH + 129): first)
  OBJ_TRY(tce_lin_rows_f32(g_mean, K, N, K, 128, w3, 0, nullptr, gh, st));
  OBJ_TRY(tce_mlp_hidden_f32(x, 0, xs, (int)N, N, din, w1, b1, w2, b2, act, gh, nullptr, partials,
                             grad, stats, st));
  return tce_out_layer_grad_f32(g_mean, h2, grad + PH, grad + PH + (int64_t)K * 128, ol_ws, N, K,
                                128, st);
}
inline int fast_bwd(const double*, int64_t, int64_t, int, int, const double*, const double*,
                    const double*, double*, double*, double*, double*, double*, int, void*) {
  tce_set_error("policy_epoch: net_kind 0 is float32 only");
  return 1;
}

// |g| of one part of a balance epoch's split gradient (grad [n], head part
// included).  Sharded: the norm of the MEAN over the ranks' gradients -- the
// part is summed over the exchange INTO `scratch` [n] (grad keeps the local
// part) and its norm scaled by grad_scale = 1 / world.
template <typename real>
int balance_norm(const real* grad, int64_t n, real* scratch, void* xchg, real grad_scale,
                 real* out, hipStream_t st) {
  typedef EpApi<real> E;
  const real* src = grad;
  real scale = 1;
  if (xchg) {
    OBJ_TRY(E::xsum_to(xchg, grad, scratch, n, (void*)st));      // (grad keeps the local part)
    src = scratch;
    scale = grad_scale;
  }
  hipLaunchKernelGGL(obj_norm_kernel<real>, dim3(1), dim3(1024), 0, st, src, n, out, scale);
  TCE_LAUNCH_CHECK();
  return 0;
}

// The end of a policy epoch behind the net's backward: Cholesky head backward,
// (gradient exchange,) clip, Adam, record row -- ONE launch where the fused tail
// applies (n <= 2^17; sharded: no clipping), else the separate launches.
// host_step: the step count including this update (the exchange's stand-alone
// Adam takes it from the host; the fused tail counts on the device).
template <typename real>
int policy_epoch_finish(real* param, real* grad, real* m, real* v, int64_t PN, int nvec, int K,
                        real* g_L, const real* gL_join, real* opt_state, unsigned* ticket,
                        const real* sur2, const real* out16, real ent_coef, real* rec_row19,
                        real lr, real beta1, real beta2, real eps, real weight_decay,
                        real clip_grad, real grad_scale, int do_adam, void* xchg, void* stream) {
  typedef EpApi<real> E;
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = PN + nvec;
  real* var = param + PN;
  real* g_var = grad + PN;
  const bool fused = do_adam && g_policy_tail_fused && n <= (1 << 17) &&
                     (!xchg || clip_grad <= real(0));
  if (fused) {
    const unsigned grid = (unsigned)tmin<int64_t>(ceil_div(n, 4 * PT_BT), PT_MAX_BLOCKS);
    XchgView X;
    if (xchg_next(xchg, n * (int64_t)sizeof(real), (int)grid, &X)) return 1;
    hipLaunchKernelGGL(policy_tail_kernel<real>, dim3(grid), dim3(PT_BT),
                       sizeof(real) * (size_t)nvec, st, param, grad, m, v, PN, nvec, K,
                       (const real*)g_L, gL_join, opt_state, ticket, sur2, out16, ent_coef,
                       rec_row19, lr, beta1, beta2, eps, weight_decay, clip_grad, grad_scale, X,
                       X.partial);
    TCE_LAUNCH_CHECK();
    return 0;
  }
  TCE_CHECK_ARG(gL_join == nullptr, "policy_epoch: internal (join pending)");
  OBJ_TRY(E::chol_bwd(var, g_L, g_var, 1, K, nvec, stream));
  if (!do_adam) return 0;                         // the caller all-reduces, steps and records
  if (xchg) {
    // (the step count lives on the device in the fused tail; here the exchange's
    // Adam wants it from the host: not reachable from policy_epoch2's arguments,
    // so this path counts on the device through tce_adam_flat after a plain sum)
    OBJ_TRY(E::xsum(xchg, grad, n, stream));
  }
  OBJ_TRY(E::adam(param, grad, m, v, n, opt_state, nullptr, lr, beta1, beta2, eps, weight_decay,
                  clip_grad, grad_scale, stream));
  return E::record(sur2, out16, opt_state + 1, ent_coef, rec_row19, stream);
}

template <typename real>
int policy_epoch2(const real* x, int64_t x_stride, int64_t N, int din, int hidden, int num_hidden,
                  int net_kind, int act, int nvec, real min_std, real* param, real* grad,
                  const real* mean_old, const real* L_old, const real* traj,
                  const real* logp_old, const real* adv, const int64_t* pairs, const real* tab,
                  int M, int nbg, real tau, real delay, real scaled_dt, real inv_scale_g,
                  int rel_goal, const real* times, int times_flags_fwd, int times_flags_bwd,
                  const real* init_time, const real* init_pos, const real* init_vel, real reg,
                  real* basis_ws, int* flag_ws, real* pair_work, real eps_mean, double eps_cov,
                  const real* beta, int entropy_eq, double* proj_ctx, real tr_coeff,
                  int tr_include_cov, real ent_coef, double* sur_ws, double* kl_ws, real* obj_ws,
                  real* ws, real* partials, real* ol_ws, int T, int P, int dof, int K, real* m,
                  real* v, real* opt_state, real lr, real beta1, real beta2, real eps,
                  real weight_decay, real clip_grad, real grad_scale, int do_adam, int balance,
                  real* rec_row19, real* bal2, void* xchg, void* stream) {
  typedef EpApi<real> E;
  TCE_CHECK_ARG(x && param && grad && ws && partials && obj_ws && rec_row19 && N > 0 &&
                    K == dof * nbg && K <= 64,
                "policy_epoch: bad arguments");
  TCE_CHECK_ARG(!xchg || do_adam, "policy_epoch: an exchange needs do_adam");
  TCE_CHECK_ARG(!do_adam || (m && v && opt_state), "policy_epoch: optimizer state missing");
  TCE_CHECK_ARG(!balance || bal2, "policy_epoch: balance needs bal2");
  const int H = hidden;
  if (net_kind == 0)
    TCE_CHECK_ARG(sizeof(real) == 4 && H == 128 && num_hidden == 2 && din >= 1 && din <= 40 &&
                      ol_ws,
                  "policy_epoch: net_kind 0 = float32 D_in <= 40 -> 128 -> 128 -> K");
  else
    TCE_CHECK_ARG(net_kind == 1 && tce_pmlp_supported(din, H, num_hidden, K, (int)sizeof(real)),
                  "policy_epoch: net shape not built (tce_pmlp_supported)");
  const int64_t PN = (int64_t)H * din + H + (num_hidden == 2 ? (int64_t)H * H + H : 0) +
                     (int64_t)K * H + K;                      // mean net parameters
  const int64_t n = PN + nvec;
  real* var = param + PN;
  real* g_var = grad + PN;
  real* h2 = ws;                                   // top hidden layer
  real* h1 = h2 + obj_up4(N * (int64_t)H);         // first hidden layer (pmlp, two layers) / gh
  real* mean = h1 + obj_up4(N * (int64_t)H);
  real* g_mean = mean + obj_up4(N * (int64_t)K);
  real* L = g_mean + obj_up4(N * (int64_t)K);
  real* g_L = L + obj_up4((int64_t)K * K);
  real* sur2 = g_L + 2 * obj_up4((int64_t)K * K);
  real* out16 = sur2 + 4;
  real* stats = out16 + 16;
  unsigned* ticket = reinterpret_cast<unsigned*>(stats + 12);   // policy_tail_kernel (zero between launches)
  real* gtmp = stats + 16;                         // balance: the surrogate's parameter gradient
  real* gtmp2 = gtmp + obj_up4(n);                 // balance, sharded: a part summed over the ranks
  hipStream_t st = (hipStream_t)stream;
  // net: forward / backward of a gradient w.r.t. the mean
  auto net_fwd = [&]() -> int {
    if (net_kind == 0) return fast_fwd(x, x_stride, N, din, act, param, h2, mean, K, stream);
    // (pmlp: h1 = first hidden layer, h2 = second; one hidden layer: h1 only)
    return E::pm_fwd(x, x_stride, N, din, H, num_hidden, K, act, param,
                     num_hidden == 2 ? h1 : h2, num_hidden == 2 ? h2 : nullptr, mean, stream);
  };
  auto net_bwd = [&](const real* gm) -> int {
    if (net_kind == 0)
      return fast_bwd(x, x_stride, N, din, act, param, h2, gm, h1, partials, ol_ws, stats, grad, K,
                      stream);
    return E::pm_bwd(x, x_stride, N, din, H, num_hidden, K, act, param,
                     num_hidden == 2 ? h1 : h2, num_hidden == 2 ? h2 : nullptr, gm, partials, grad,
                     stream);
  };
  // ---- forward
  OBJ_TRY(E::begin(var, nvec, min_std, L_old, eps_cov, beta, entropy_eq, proj_ctx, L, obj_ws, N, K,
                   P, stream));
  OBJ_TRY(net_fwd());
  OBJ_TRY(E::objective(mean, L, mean_old, L_old, traj, logp_old, adv, pairs, tab, M, nbg, tau,
                       delay, scaled_dt, inv_scale_g, rel_goal, times, times_flags_fwd,
                       times_flags_bwd, init_time, init_pos, init_vel, reg, basis_ws, flag_ws,
                       pair_work, eps_mean, eps_cov, beta, entropy_eq, proj_ctx, tr_coeff,
                       tr_include_cov, ent_coef, sur_ws, kl_ws, obj_ws, g_mean, g_L, sur2, out16, N,
                       T, P, dof, K, 1, balance ? 3 : 1, stream));
  const bool fused_tail = do_adam && g_policy_tail_fused && n <= (1 << 17) &&
                          (!xchg || clip_grad <= real(0));
  const real* gL_join = nullptr;                   // fused tail: the half of d / d L still to add
  if (!balance) {
    // ---- backward into the flat gradient
    OBJ_TRY(net_bwd(g_mean));
    if (fused_tail) {
      // the join without its add (policy_objective_end): the tail kernel adds
      const bool single = g_obj_streams < 2;
      ObjSide* S = single ? nullptr : obj_side();
      TCE_CHECK_ARG(single || S != nullptr, "policy_objective: could not create the side stream");
      OBJ_HIP(hipStreamWaitEvent(st, S->ev[4], 0));
      gL_join = obj_proj_L(obj_ws, N, K, P) + 2 * obj_up4((int64_t)K * K);
    } else {
      OBJ_TRY(policy_objective_end<real>(g_L, obj_ws, N, K, P, st));
    }
  } else {
    const bool single = g_obj_streams < 2;
    ObjSide* S = single ? nullptr : obj_side();
    TCE_CHECK_ARG(single || S != nullptr, "policy_objective: could not create the side stream");
    const int64_t KK = obj_up4((int64_t)K * K);
    real* gm_p = obj_ws + 2 * obj_up4(N * (int64_t)K);      // surrogate: d / d mean_new
    real* gL_p = obj_proj_L(obj_ws, N, K, P) + 2 * KK;      //            d / d L_new
    real* gL_e = gL_p + KK;                                 // entropy term: d / d L_new
    // surrogate alone
    OBJ_TRY(net_bwd(gm_p));
    OBJ_HIP(hipStreamWaitEvent(st, S->ev[4], 0));
    OBJ_TRY(E::chol_bwd(var, gL_p, g_var, 1, K, nvec, stream));
    OBJ_TRY(balance_norm<real>(grad, n, gtmp2, xchg, grad_scale, bal2, st));
    OBJ_HIP_ALWAYS(hipMemcpyAsync(gtmp, grad, sizeof(real) * PN, hipMemcpyDeviceToDevice, st));
    // trust region loss alone
    OBJ_TRY(net_bwd(g_mean));
    OBJ_TRY(E::chol_bwd(var, g_L, g_var, 1, K, nvec, stream));
    OBJ_TRY(balance_norm<real>(grad, n, gtmp2, xchg, grad_scale, bal2 + 1, st));
    // the epoch's gradient: their sum (+ the entropy term's way through the factor)
    hipLaunchKernelGGL(obj_add3_kernel<real>, dim3((unsigned)ceil_div(PN, 256)), dim3(256), 0, st,
                       grad, gtmp, (const real*)nullptr, PN);
    TCE_LAUNCH_CHECK();
    hipLaunchKernelGGL(obj_add3_kernel<real>, dim3((unsigned)ceil_div((int64_t)K * K, 256)),
                       dim3(256), 0, st, g_L, gL_p,
                       ent_coef != real(0) ? (const real*)gL_e : (const real*)nullptr,
                       (int64_t)K * K);
    TCE_LAUNCH_CHECK();
  }
  return policy_epoch_finish<real>(param, grad, m, v, PN, nvec, K, g_L, gL_join, opt_state, ticket,
                                   sur2, out16, ent_coef, rec_row19, lr, beta1, beta2, eps,
                                   weight_decay, clip_grad, grad_scale, do_adam, xchg, stream);
}


// ---------------------------------------------------------------------------
// One whole policy epoch of the black-box agent (black_box_agent.py:225-339)
// without autograd in ONE call, for the mean nets the row kernels of
// csrc/smlp.hip do not cover -- box pushing's 128 x 2 and table tennis's 256 x 1
// (mprl/config/box_push_random_init/bbrl/entire/shared.yaml:66-67,
// mprl/config/table_tennis_4d/bbrl/entire/shared.yaml:72-73): Cholesky head +
// covariance projection on the second stream beside the mean net's forward, the
// black-box objective and its gradient (bb_policy_objective), the mean net's
// backward into the flat gradient, then policy_epoch2's tail (Cholesky head
// backward, clip, Adam, record row).  net_kind / balance / do_adam as there; the
// workspace has policy_epoch2's layout, obj_ws the one of
// tce_bb_policy_objective_*.  mean / L of this epoch stay in ws (mean at
// 2 up4(N H), L behind mean and g_mean); proj_*_out (nullable) receive the
// projected distribution.
// ---------------------------------------------------------------------------
template <typename real>
int bb_policy_epoch(const real* x, int64_t x_stride, int64_t N, int din, int hidden, int num_hidden,
                    int net_kind, int act, int nvec, real min_std, real* param, real* grad,
                    const real* mean_old, const real* L_old, const real* actions,
                    const real* logp_old, const real* adv, real eps_mean, double eps_cov,
                    const real* beta, int entropy_eq, double* proj_ctx, real tr_coeff,
                    int tr_include_cov, real ent_coef, double* sur_ws, double* kl_ws, real* obj_ws,
                    real* ws, real* partials, real* ol_ws, int K, real* m, real* v,
                    real* opt_state, real lr, real beta1, real beta2, real eps, real weight_decay,
                    real clip_grad, real grad_scale, int do_adam, int balance, real* rec_row19,
                    real* bal2, real* proj_mean_out, real* proj_L_out, void* xchg, void* stream) {
  typedef EpApi<real> E;
  typedef ObjApi<real> A;
  TCE_CHECK_ARG(!xchg || do_adam, "bb_policy_epoch: an exchange needs do_adam");
  TCE_CHECK_ARG(x && param && grad && ws && partials && obj_ws && rec_row19 && N > 0 && K > 0 &&
                    K <= 64 && mean_old && L_old && proj_ctx,
                "bb_policy_epoch: bad arguments");
  TCE_CHECK_ARG(!do_adam || (m && v && opt_state), "bb_policy_epoch: optimizer state missing");
  TCE_CHECK_ARG(!balance || bal2, "bb_policy_epoch: balance needs bal2");
  const int H = hidden;
  if (net_kind == 0)
    TCE_CHECK_ARG(sizeof(real) == 4 && H == 128 && num_hidden == 2 && din >= 1 && din <= 40 &&
                      ol_ws,
                  "bb_policy_epoch: net_kind 0 = float32 D_in <= 40 -> 128 -> 128 -> K");
  else
    TCE_CHECK_ARG(net_kind == 1 && tce_pmlp_supported(din, H, num_hidden, K, (int)sizeof(real)),
                  "bb_policy_epoch: net shape not built (tce_pmlp_supported)");
  const int64_t PN = (int64_t)H * din + H + (num_hidden == 2 ? (int64_t)H * H + H : 0) +
                     (int64_t)K * H + K;
  const int64_t n = PN + nvec;
  const int64_t KK = (int64_t)K * K;
  real* var = param + PN;
  real* g_var = grad + PN;
  real* h2 = ws;
  real* h1 = h2 + obj_up4(N * (int64_t)H);
  real* mean = h1 + obj_up4(N * (int64_t)H);
  real* g_mean = mean + obj_up4(N * (int64_t)K);
  real* L = g_mean + obj_up4(N * (int64_t)K);
  real* g_L = L + obj_up4(KK);
  real* sur2 = g_L + 2 * obj_up4(KK);
  real* out16 = sur2 + 4;
  real* stats = out16 + 16;
  unsigned* ticket = reinterpret_cast<unsigned*>(stats + 12);
  real* gtmp = stats + 16;
  real* gtmp2 = gtmp + obj_up4(n);
  hipStream_t st = (hipStream_t)stream;
  const bool single = g_obj_streams < 2;
  ObjSide* S = single ? nullptr : obj_side();
  TCE_CHECK_ARG(single || S != nullptr, "policy_objective: could not create the side stream");
  hipStream_t sd = single ? st : S->side;
  // bb_policy_objective's workspace: where the projected factor and the parts of
  // a split gradient live
  real* gm_p = obj_ws + 2 * obj_up4(N * (int64_t)K);
  real* pL = obj_ws + 3 * obj_up4(N * (int64_t)K) + 2 * obj_up4(N);
  real* gL_p = pL + 2 * obj_up4(KK);
  real* gL_e = gL_p + obj_up4(KK);
  auto net_fwd = [&]() -> int {
    if (net_kind == 0) return fast_fwd(x, x_stride, N, din, act, param, h2, mean, K, stream);
    return E::pm_fwd(x, x_stride, N, din, H, num_hidden, K, act, param,
                     num_hidden == 2 ? h1 : h2, num_hidden == 2 ? h2 : nullptr, mean, stream);
  };
  auto net_bwd = [&](const real* gm) -> int {
    if (net_kind == 0)
      return fast_bwd(x, x_stride, N, din, act, param, h2, gm, h1, partials, ol_ws, stats, grad, K,
                      stream);
    return E::pm_bwd(x, x_stride, N, din, H, num_hidden, K, act, param,
                     num_hidden == 2 ? h1 : h2, num_hidden == 2 ? h2 : nullptr, gm, partials, grad,
                     stream);
  };
  // ---- forward: head + covariance projection beside the mean net
  OBJ_HIP(hipEventRecord(S->ev[0], st));
  OBJ_HIP(hipStreamWaitEvent(sd, S->ev[0], 0));
  OBJ_TRY(A::chol_fwd(var, L, 1, K, nvec, min_std, sd));
  OBJ_TRY(A::proj_fwd(L, L_old, 0, eps_cov, beta, entropy_eq, pL, proj_ctx, 1, K, 1, sd));
  OBJ_HIP(hipEventRecord(S->ev[1], sd));
  OBJ_TRY(net_fwd());
  OBJ_TRY(bb_policy_objective<real>(mean, L, mean_old, L_old, actions, logp_old, adv, eps_mean,
                                    eps_cov, beta, entropy_eq, proj_ctx, tr_coeff, tr_include_cov,
                                    ent_coef, sur_ws, kl_ws, obj_ws, g_mean, g_L, sur2, out16,
                                    proj_mean_out, proj_L_out, N, K, st, 1, balance ? 1 : 0));
  if (!balance) {
    OBJ_TRY(net_bwd(g_mean));
  } else {
    // surrogate alone
    OBJ_TRY(net_bwd(gm_p));
    OBJ_TRY(E::chol_bwd(var, gL_p, g_var, 1, K, nvec, stream));
    OBJ_TRY(balance_norm<real>(grad, n, gtmp2, xchg, grad_scale, bal2, st));
    OBJ_HIP_ALWAYS(hipMemcpyAsync(gtmp, grad, sizeof(real) * PN, hipMemcpyDeviceToDevice, st));
    // trust region loss alone
    OBJ_TRY(net_bwd(g_mean));
    OBJ_TRY(E::chol_bwd(var, g_L, g_var, 1, K, nvec, stream));
    OBJ_TRY(balance_norm<real>(grad, n, gtmp2, xchg, grad_scale, bal2 + 1, st));
    // the epoch's gradient: their sum (+ the entropy term's way through the factor)
    hipLaunchKernelGGL(obj_add3_kernel<real>, dim3((unsigned)ceil_div(PN, 256)), dim3(256), 0, st,
                       grad, gtmp, (const real*)nullptr, PN);
    TCE_LAUNCH_CHECK();
    hipLaunchKernelGGL(obj_add3_kernel<real>, dim3((unsigned)ceil_div(KK, 256)), dim3(256), 0, st,
                       g_L, gL_p, ent_coef != real(0) ? (const real*)gL_e : (const real*)nullptr,
                       KK);
    TCE_LAUNCH_CHECK();
  }
  return policy_epoch_finish<real>(param, grad, m, v, PN, nvec, K, g_L, (const real*)nullptr,
                                   opt_state, ticket, sur2, out16, ent_coef, rec_row19, lr, beta1,
                                   beta2, eps, weight_decay, clip_grad, grad_scale, do_adam, xchg,
                                   stream);
}

}  // namespace

extern "C" {

int64_t tce_kl_shared_ws_len(int64_t N) { return 3 * ceil_div(N, KE_EPB); }

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

int64_t tce_policy_objective_ws_len(int64_t N, int K, int P) { return obj_ws_len(N, K, P); }

int64_t tce_bb_policy_objective_ws_len(int64_t N, int K) { return bb_obj_ws_len(N, K); }

int tce_policy_objective_streams(int n) {
  g_obj_streams = n < 2 ? 1 : 2;
  return 0;
}

int tce_policy_objective_use_stream(void* stream) {
  TCE_CHECK_ARG(stream != nullptr, "policy_objective_use_stream: null stream");
  g_obj_given = (hipStream_t)stream;
  return 0;
}

int tce_policy_objective_side_stream(void** stream) {
  TCE_CHECK_ARG(stream != nullptr, "policy_objective_side_stream: null output");
  ObjSide* S = obj_side();
  TCE_CHECK_ARG(S != nullptr, "policy_objective: could not create the side stream");
  *stream = reinterpret_cast<void*>(S->side);
  return 0;
}

#define DEFINE_POLICY_OBJECTIVE(SFX, REAL)                                                \
  int tce_policy_objective_##SFX(                                                         \
      const REAL* mean_new, const REAL* L_new, const REAL* mean_old, const REAL* L_old,   \
      const REAL* traj, const REAL* logp_old, const REAL* adv, const int64_t* pairs,      \
      const REAL* tab, int M, int nbg, REAL tau, REAL delay, REAL scaled_dt,              \
      REAL inv_scale_g, int rel_goal, const REAL* times, int times_flags_fwd,             \
      int times_flags_bwd, const REAL* init_time, const REAL* init_pos,                   \
      const REAL* init_vel, REAL reg, REAL* basis_ws, int* flag_ws, REAL* pair_work,      \
      REAL eps_mean, double eps_cov, const REAL* beta, int entropy_eq, double* proj_ctx,  \
      REAL tr_coeff, int tr_include_cov, REAL ent_coef, double* sur_ws, double* kl_ws,    \
      REAL* ws, REAL* grad_mean, REAL* grad_L, REAL* sur2, REAL* out16, int64_t N, int T, \
      int P, int dof, int K, int proj_started, int defer_join, void* stream) {                                              \
    return policy_objective<REAL>(                                                        \
        mean_new, L_new, mean_old, L_old, traj, logp_old, adv, pairs, tab, M, nbg, tau,   \
        delay, scaled_dt, inv_scale_g, rel_goal, times, times_flags_fwd, times_flags_bwd, \
        init_time, init_pos, init_vel, reg, basis_ws, flag_ws, pair_work, eps_mean,       \
        eps_cov, beta, entropy_eq, proj_ctx, tr_coeff, tr_include_cov, ent_coef, sur_ws,  \
        kl_ws, ws, grad_mean, grad_L, sur2, out16, N, T, P, dof, K, proj_started,         \
        defer_join, (hipStream_t)stream);                                                 \
  }                                                                                       \
  int tce_policy_objective_end_##SFX(REAL* grad_L, REAL* ws, int64_t N, int K, int P,     \
                                     void* stream) {                                      \
    return policy_objective_end<REAL>(grad_L, ws, N, K, P, (hipStream_t)stream);          \
  }                                                                                       \
  int tce_policy_objective_begin_##SFX(                                                   \
      const REAL* var_vec, int nvec, REAL min_std, const REAL* L_old, double eps_cov,     \
      const REAL* beta, int entropy_eq, double* proj_ctx, REAL* L_new, REAL* ws,          \
      int64_t N, int K, int P, void* stream) {                                            \
    return policy_objective_begin<REAL>(var_vec, nvec, min_std, L_old, eps_cov, beta,     \
                                        entropy_eq, proj_ctx, L_new, ws, N, K, P,         \
                                        (hipStream_t)stream);                             \
  }
#define DEFINE_BB_POLICY_OBJECTIVE(SFX, REAL)                                             \
  int tce_bb_policy_objective_##SFX(                                                      \
      const REAL* mean_new, const REAL* L_new, const REAL* mean_old, const REAL* L_old,   \
      const REAL* actions, const REAL* logp_old, const REAL* adv, REAL eps_mean,          \
      double eps_cov, const REAL* beta, int entropy_eq, double* proj_ctx, REAL tr_coeff,  \
      int tr_include_cov, REAL ent_coef, double* sur_ws, double* kl_ws, REAL* ws,         \
      REAL* grad_mean, REAL* grad_L, REAL* sur2, REAL* out16, REAL* proj_mean_out,        \
      REAL* proj_L_out, int64_t N, int K, void* stream) {                                 \
    return bb_policy_objective<REAL>(                                                     \
        mean_new, L_new, mean_old, L_old, actions, logp_old, adv, eps_mean, eps_cov,      \
        beta, entropy_eq, proj_ctx, tr_coeff, tr_include_cov, ent_coef, sur_ws, kl_ws,    \
        ws, grad_mean, grad_L, sur2, out16, proj_mean_out, proj_L_out, N, K,              \
        (hipStream_t)stream);                                                             \
  }
DEFINE_BB_POLICY_OBJECTIVE(f32, float)
DEFINE_BB_POLICY_OBJECTIVE(f64, double)
DEFINE_POLICY_OBJECTIVE(f32, float)
DEFINE_POLICY_OBJECTIVE(f64, double)

int64_t tce_out_layer_grad_ws_len(int64_t N, int K, int H) {
  return ceil_div(N, OL_ROWS) * ((int64_t)K * H + K);
}
int tce_out_layer_grad_f32(const float* grad_out, const float* hidden, float* grad_W,
                           float* grad_b, float* ws, int64_t N, int K, int H, void* stream) {
  return out_layer_grad<float>(grad_out, hidden, grad_W, grad_b, ws, N, K, H,
                               (hipStream_t)stream);
}
int tce_out_layer_grad_f64(const double* grad_out, const double* hidden, double* grad_W,
                           double* grad_b, double* ws, int64_t N, int K, int H, void* stream) {
  return out_layer_grad<double>(grad_out, hidden, grad_W, grad_b, ws, N, K, H,
                                (hipStream_t)stream);
}
// The K x K half of tce_kl_shared alone: the caller supplies the sums of the
// three Mahalanobis terms (partials: double [nparts][3], added in order).
#define DEFINE_KL_SHARED_MAT(SFX, REAL)                                                     \
  int tce_kl_shared_mat_##SFX(const REAL* L_new, const REAL* L_old, const REAL* L_proj,      \
                              int64_t N, int K, REAL tr_coeff, int tr_include_cov,          \
                              const double* partials, int nparts, REAL* out16,              \
                              REAL* grad_L, void* stream) {                                 \
    TCE_CHECK_ARG(L_new && L_old && L_proj && partials && out16 && N > 0 && K > 0 &&        \
                      K <= 64 && nparts > 0,                                                \
                  "kl_shared_mat: bad arguments (K <= 64)");                                \
    const int par = (size_t)6 * K * sm_pitch(K) * sizeof(double) <= 150 * 1024;             \
    const size_t lds_m = (size_t)(par ? 6 : 4) * K * sm_pitch(K) * sizeof(double);          \
    if (lds_m > 48 * 1024)                                                                  \
      tce_lds_limit(reinterpret_cast<const void*>(kl_shared_mat_kernel<REAL>), lds_m);      \
    hipLaunchKernelGGL(kl_shared_mat_kernel<REAL>, dim3(1), dim3(SM_BT), lds_m,             \
                       (hipStream_t)stream, L_new, L_old, L_proj, N, K, tr_coeff,           \
                       tr_include_cov, partials, nparts, out16, grad_L, par);               \
    TCE_LAUNCH_CHECK();                                                                     \
    return 0;                                                                               \
  }
DEFINE_KL_SHARED_MAT(f32, float)
DEFINE_KL_SHARED_MAT(f64, double)

// ---------------------------------------------------------------------------
// One whole TCE policy epoch without autograd in ONE call (what
// rl/objective.py:DirectEpoch.run issued as 12 separate calls: the epoch was
// host-bound at 25 - 35 us per call): Cholesky head + covariance projection on
// the second stream, mean net forward (two 128-wide hidden layers on the fused
// MFMA kernel, output layer on the row kernel), the objective and its gradient
// (deferred join), output layer input gradient, hidden layers' backward into
// the flat gradient, output layer weight gradient, join, Cholesky head
// backward, flat Adam (do_adam; a sharded caller all-reduces first), record.
// Parameters FLAT in the order W1 [128][din] | b1 | W2 [128][128] | b2 | W3
// [K][128] | b3 | variance vector [nvec] (the FlatAdam buffer of the policy).
// ws: float [tce_policy_epoch_ws_len(N, K)].
int64_t tce_policy_epoch_ws_len(int64_t N, int K) {
  return tce_policy_epoch2_ws_len(N, K, 128, 0);
}

int tce_policy_epoch_f32(
    const float* x, int64_t x_stride, int64_t N, int din, int act, int nvec, float min_std,
    float* param, float* grad, const float* mean_old, const float* L_old, const float* traj,
    const float* logp_old, const float* adv, const int64_t* pairs, const float* tab, int M, int nbg,
    float tau, float delay, float scaled_dt, float inv_scale_g, int rel_goal, const float* times,
    int times_flags_fwd, int times_flags_bwd, const float* init_time, const float* init_pos,
    const float* init_vel, float reg, float* basis_ws, int* flag_ws, float* pair_work,
    float eps_mean, double eps_cov, const float* beta, int entropy_eq, double* proj_ctx,
    float tr_coeff, int tr_include_cov, float ent_coef, double* sur_ws, double* kl_ws,
    float* obj_ws, float* ws, float* partials, float* ol_ws, int T, int P, int dof, int K,
    float* m, float* v, float* opt_state, float lr, float beta1, float beta2, float eps,
    float weight_decay, float clip_grad, float grad_scale, int do_adam, float* rec_row19,
    void* stream) {
  return tce_policy_epoch2_f32(
      x, x_stride, N, din, 128, 2, 0, act, nvec, min_std, param, grad, mean_old, L_old, traj,
      logp_old, adv, pairs, tab, M, nbg, tau, delay, scaled_dt, inv_scale_g, rel_goal, times,
      times_flags_fwd, times_flags_bwd, init_time, init_pos, init_vel, reg, basis_ws, flag_ws,
      pair_work, eps_mean, eps_cov, beta, entropy_eq, proj_ctx, tr_coeff, tr_include_cov, ent_coef,
      sur_ws, kl_ws, obj_ws, ws, partials, ol_ws, T, P, dof, K, m, v, opt_state, lr, beta1, beta2,
      eps, weight_decay, clip_grad, grad_scale, do_adam, 0, rec_row19, nullptr, nullptr, stream);
}

void tce_policy_tail_fused(int on) { g_policy_tail_fused = on; }
void tce_policy_inline_surrogate(int on) { g_inline_surrogate = on; }

int64_t tce_policy_epoch2_ws_len(int64_t N, int K, int hidden, int64_t nparam) {
  return epoch2_ws_len(N, K, hidden, nparam);
}

#define DEFINE_POLICY_EPOCH2(SFX, REAL)                                                      \
  int tce_policy_epoch2_##SFX(                                                               \
      const REAL* x, int64_t x_stride, int64_t N, int din, int hidden, int num_hidden,       \
      int net_kind, int act, int nvec, REAL min_std, REAL* param, REAL* grad,                \
      const REAL* mean_old, const REAL* L_old, const REAL* traj, const REAL* logp_old,       \
      const REAL* adv, const int64_t* pairs, const REAL* tab, int M, int nbg, REAL tau,      \
      REAL delay, REAL scaled_dt, REAL inv_scale_g, int rel_goal, const REAL* times,         \
      int times_flags_fwd, int times_flags_bwd, const REAL* init_time, const REAL* init_pos, \
      const REAL* init_vel, REAL reg, REAL* basis_ws, int* flag_ws, REAL* pair_work,         \
      REAL eps_mean, double eps_cov, const REAL* beta, int entropy_eq, double* proj_ctx,     \
      REAL tr_coeff, int tr_include_cov, REAL ent_coef, double* sur_ws, double* kl_ws,       \
      REAL* obj_ws, REAL* ws, REAL* partials, REAL* ol_ws, int T, int P, int dof, int K,     \
      REAL* m, REAL* v, REAL* opt_state, REAL lr, REAL beta1, REAL beta2, REAL eps,          \
      REAL weight_decay, REAL clip_grad, REAL grad_scale, int do_adam, int balance,          \
      REAL* rec_row19, REAL* bal2, void* xchg, void* stream) {                               \
    return policy_epoch2<REAL>(                                                              \
        x, x_stride, N, din, hidden, num_hidden, net_kind, act, nvec, min_std, param, grad,  \
        mean_old, L_old, traj, logp_old, adv, pairs, tab, M, nbg, tau, delay, scaled_dt,     \
        inv_scale_g, rel_goal, times, times_flags_fwd, times_flags_bwd, init_time, init_pos, \
        init_vel, reg, basis_ws, flag_ws, pair_work, eps_mean, eps_cov, beta, entropy_eq,    \
        proj_ctx, tr_coeff, tr_include_cov, ent_coef, sur_ws, kl_ws, obj_ws, ws, partials,   \
        ol_ws, T, P, dof, K, m, v, opt_state, lr, beta1, beta2, eps, weight_decay,           \
        clip_grad, grad_scale, do_adam, balance, rec_row19, bal2, xchg, stream);             \
  }
DEFINE_POLICY_EPOCH2(f32, float)
DEFINE_POLICY_EPOCH2(f64, double)

#define DEFINE_BB_POLICY_EPOCH(SFX, REAL)                                                    \
  int tce_bb_policy_epoch_##SFX(                                                             \
      const REAL* x, int64_t x_stride, int64_t N, int din, int hidden, int num_hidden,       \
      int net_kind, int act, int nvec, REAL min_std, REAL* param, REAL* grad,                \
      const REAL* mean_old, const REAL* L_old, const REAL* actions, const REAL* logp_old,    \
      const REAL* adv, REAL eps_mean, double eps_cov, const REAL* beta, int entropy_eq,      \
      double* proj_ctx, REAL tr_coeff, int tr_include_cov, REAL ent_coef, double* sur_ws,    \
      double* kl_ws, REAL* obj_ws, REAL* ws, REAL* partials, REAL* ol_ws, int K, REAL* m,    \
      REAL* v, REAL* opt_state, REAL lr, REAL beta1, REAL beta2, REAL eps,                   \
      REAL weight_decay, REAL clip_grad, REAL grad_scale, int do_adam, int balance,          \
      REAL* rec_row19, REAL* bal2, REAL* proj_mean_out, REAL* proj_L_out, void* xchg,        \
      void* stream) {                                                                        \
    return bb_policy_epoch<REAL>(                                                            \
        x, x_stride, N, din, hidden, num_hidden, net_kind, act, nvec, min_std, param, grad,  \
        mean_old, L_old, actions, logp_old, adv, eps_mean, eps_cov, beta, entropy_eq,        \
        proj_ctx, tr_coeff, tr_include_cov, ent_coef, sur_ws, kl_ws, obj_ws, ws, partials,   \
        ol_ws, K, m, v, opt_state, lr, beta1, beta2, eps, weight_decay, clip_grad,           \
        grad_scale, do_adam, balance, rec_row19, bal2, proj_mean_out, proj_L_out, xchg,      \
        stream);                                                                             \
  }
DEFINE_BB_POLICY_EPOCH(f32, float)
DEFINE_BB_POLICY_EPOCH(f64, double)

int tce_policy_record_f32(const float* sur2, const float* out16, const float* norms2,
                          float ent_coef, float* row19, void* stream) {
  TCE_CHECK_ARG(sur2 && out16 && norms2 && row19, "policy_record: null buffer");
  hipLaunchKernelGGL(policy_record_kernel<float>, dim3(1), dim3(64), 0, (hipStream_t)stream,
                     sur2, out16, norms2, ent_coef, row19);
  TCE_LAUNCH_CHECK();
  return 0;
}
int tce_policy_record_f64(const double* sur2, const double* out16, const double* norms2,
                          double ent_coef, double* row19, void* stream) {
  TCE_CHECK_ARG(sur2 && out16 && norms2 && row19, "policy_record: null buffer");
  hipLaunchKernelGGL(policy_record_kernel<double>, dim3(1), dim3(64), 0, (hipStream_t)stream,
                     sur2, out16, norms2, ent_coef, row19);
  TCE_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
