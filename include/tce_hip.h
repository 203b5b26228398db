/* libtce_hip.so -- C ABI of the MI355X (gfx950) TCE/BBRL hot path.
 *
 * The reference (BruceGeLi/TCE_RL) has no FFI: its boundary is a set of Python
 * classes.  This header is the C-ABI layer the build puts underneath those
 * classes; every entry point names the reference code it replaces (paths are
 * relative to /root/reference/).  Conventions:
 *   - all pointers are DEVICE pointers (HIP), row-major, contiguous unless a
 *     stride argument says otherwise; bool tensors are 1 byte per element;
 *   - `stream` is a hipStream_t (NULL = default stream); calls only enqueue;
 *   - return 0 on success, non-zero on error (message: tce_last_error());
 *     nothing aborts the process;
 *   - *_f32 / *_f64 are the two dtypes the reference accepts
 *     (mprl/util/util_data_structure.py:59-78);
 *   - the library keeps no pointers after a call returns.
 */
#ifndef TCE_HIP_H
#define TCE_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library ---------------------------------------------------------- */
const char* tce_last_error(void);
int tce_version(void);
int tce_device_count(void);

/* ---- GAE + segment advantage ------------------------------------------
 * TemporalCorrelatedAgent.get_advantage_return
 *   (mprl/rl/agent/temporal_correlated_agent.py:118-181) and, when P > 0, the
 * un-normalised `value_subtraction` segment advantage (:236-279) fused in.
 * rewards/dones/tl_dones/adv/ret [N,T]; values [N,T+1]; pairs int64 [P,2];
 * seg_out [N,P]; partials double [tce_gae_num_partials(N), 3] =
 * per-workgroup (count, mean, M2) of seg_out for the global normalisation.
 * Results are bit-identical to the reference's op order (no FMA contraction).
 */
int64_t tce_gae_num_partials(int64_t N);
int tce_gae_f32(const float* rewards, const float* values, const uint8_t* dones,
                const uint8_t* tl_dones, float* adv, float* ret,
                const int64_t* pairs, int P, float* seg_out, double* partials,
                int64_t N, int T, float gamma, float lam, int use_gae,
                void* stream);
int tce_gae_f64(const double* rewards, const double* values, const uint8_t* dones,
                const uint8_t* tl_dones, double* adv, double* ret,
                const int64_t* pairs, int P, double* seg_out, double* partials,
                int64_t N, int T, double gamma, double lam, int use_gae,
                void* stream);

/* ---- global mean / unbiased std (advantage normalisation) ---------------
 * `(x - x.mean()) / (x.std() + 1e-8)` of temporal_correlated_agent.py:213-215,
 * 230-233,281-284 and black_box_agent.py:93-97, split so that multi-GPU runs
 * can merge the (count, mean, M2) triples of all ranks before normalising.
 * partials: double [tce_moments_num_partials(), 3]; stats: double[3] =
 * {count, mean, M2}.  tce_normalize: y = clamp((x-mean)/(std+eps), +-clip);
 * stats == NULL -> clamp only; clip <= 0 -> no clamp; single_std_one: std := 1
 * when count == 1 (black_box_agent.py:95).
 */
int64_t tce_moments_num_partials(void);
int tce_moments_partial_f32(const float* x, int64_t n, double* partials, void* stream);
int tce_moments_partial_f64(const double* x, int64_t n, double* partials, void* stream);
int tce_moments_finalize(const double* partials, int nparts, double* stats, void* stream);
int tce_normalize_f32(const float* x, float* y, int64_t n, const double* stats,
                      float eps, float clip, int single_std_one, void* stream);
int tce_normalize_f64(const double* x, double* y, int64_t n, const double* stats,
                      double eps, double clip, int single_std_one, void* stream);

/* ---- other segment-advantage modes -------------------------------------
 * `accumulate` (temporal_correlated_agent.py:211-228): out[n,p] =
 * sum_{t=a..b inclusive} f(adv[n,t]), f = optional normalise (stats) + clamp.
 * `accumulated_rewards` (:288-319): (sum_{[a,b)} gamma^t r - column mean)/gamma^a.
 * col_mean: NULL = the column mean over these N rows (the reference's
 * accumulated_rewards.mean(dim=0)); else P means of the GLOBAL batch supplied
 * by the caller (env shards: all-reduced column sums / global N).  center = 0
 * returns the raw sums (first pass of the sharded case).
 */
int tce_segment_accumulate_f32(const float* adv, const int64_t* pairs, int P,
                               float* out, int64_t N, int T, const double* stats,
                               float eps, float clip, void* stream);
int tce_segment_accumulate_f64(const double* adv, const int64_t* pairs, int P,
                               double* out, int64_t N, int T, const double* stats,
                               double eps, double clip, void* stream);
int tce_segment_accrew_f32(const float* rewards, const int64_t* pairs, int P,
                           float* out, int64_t N, int T, float gamma,
                           const float* col_mean, int center, void* stream);
int tce_segment_accrew_f64(const double* rewards, const int64_t* pairs, int P,
                           double* out, int64_t N, int T, double gamma,
                           const double* col_mean, int center, void* stream);

/* ---- ProDMP trajectory generator ---------------------------------------
 * The arithmetic of mp_pytorch==0.1.4 (third-party, un-vendored; pinned at
 * conda_env.sh:41) behind the reference's call sites
 *   mprl/util/util_mp.py:11-46 (constructor surface),
 *   mprl/rl/policy/temporal_correlated_policy.py:74-102 (sample),
 *   mprl/rl/sampler/temporal_correlated_sampler.py:64-78 (time grid).
 * `tab` [M, 4 + 2*nbg]: host pre-computed table (y1, y2, dy1, dy2, scaled
 * position basis, scaled velocity basis) on a grid of step `scaled_dt` in
 * scaled time; nbg = num_basis + 1 <= 16; dof <= 8.
 * times [N,T]; times_flags bit 0 set: rows may differ arbitrarily (per-element
 * basis evaluation); otherwise the rows are the sampler's affine grid and the
 * kernels share one basis table when all init_time are equal (checked on the
 * device, no host sync).  basis_ws: real [T, 4 + 2*nbg] workspace, flag_ws:
 * int[1] workspace; times_flags bit 1 set: both already hold the table of
 * exactly these times / init_time (left there by an earlier call: the table is
 * not rebuilt -- one rollout or update evaluates the same grid ~100 times);
 * pair log-prob only, bit 2 set: `work` still holds the per-pair factors of
 * this very L from the forward call (the backward then skips that kernel);
 * bit 3 set (with bit 0 clear, shared L): the caller has checked on the host
 * that all init_time are equal, so the general-path kernels -- which exit at
 * once on the device in that case -- are not launched at all.
 * out [N, T, 2*dof] = cat[pos, vel].
 */
int tce_times_f32(const float* init_time, float off_first, float off_last,
                  float* times, int64_t N, int T, void* stream);
int tce_times_f64(const double* init_time, double off_first, double off_last,
                  double* times, int64_t N, int T, void* stream);
/* params = mean + L eps (MultivariateNormal(loc, scale_tril).rsample with the
 * noise passed in; black_box_policy.py:80-93).  L_stride: elements between
 * consecutive matrices, 0 = one shared [K,K] matrix. */
int tce_mvn_rsample_f32(const float* mean, const float* L, int64_t L_stride,
                        const float* eps, float* out, int64_t N, int K, void* stream);
int tce_mvn_rsample_f64(const double* mean, const double* L, int64_t L_stride,
                        const double* eps, double* out, int64_t N, int K, void* stream);
int tce_prodmp_traj_f32(const float* tab, int M, int nbg, float tau, float delay,
                        float scaled_dt, float inv_scale_g, int rel_goal,
                        const float* times, int times_flags, const float* params,
                        const float* init_time, const float* init_pos,
                        const float* init_vel, float* out, float* basis_ws,
                        int* flag_ws, int64_t N, int T, int dof, void* stream);
int tce_prodmp_traj_f64(const double* tab, int M, int nbg, double tau, double delay,
                        double scaled_dt, double inv_scale_g, int rel_goal,
                        const double* times, int times_flags, const double* params,
                        const double* init_time, const double* init_pos,
                        const double* init_vel, double* out, double* basis_ws,
                        int* flag_ws, int64_t N, int T, int dof, void* stream);

/* ---- pair-wise trajectory log-probability -------------------------------
 * TemporalCorrelatedPolicy.log_prob
 *   (mprl/rl/policy/temporal_correlated_policy.py:104-203): logp [N,P] of the
 * 2*dof positions at the two times of every pair under N(H theta + c,
 * H L L^T H^T + reg I).  traj [N,T,2*dof], mean [N,K], L [.,K,K] with
 * L_stride (0 = shared), pairs int64 [P,2].  bwd: grad_mean [N,K] and
 * grad_L ([N,K,K] per env, or [K,K] when L_stride == 0; lower triangle) for
 * grad_logp [N,P]; the forward is recomputed, nothing is saved between the two
 * calls.  work: scratch of tce_pair_logprob_work_len(...) elements (NULL when
 * that is 0, i.e. for a per-env L).  With a shared L and a common init time
 * the env-independent factorisations are done once per pair.
 */
int64_t tce_pair_logprob_work_len(int64_t N, int P, int dof, int nbg, int64_t L_stride,
                                  int bwd);
int tce_pair_logprob_fwd_f32(
    const float* traj, const float* mean, const float* L, int64_t L_stride,
    const int64_t* pairs, const float* tab, int M, int nbg, float tau, float delay,
    float scaled_dt, float inv_scale_g, int rel_goal, const float* times,
    int times_flags, const float* init_time, const float* init_pos,
    const float* init_vel, float reg, float* logp, float* basis_ws, int* flag_ws,
    float* work, int64_t N, int T, int P, int dof, void* stream);
int tce_pair_logprob_fwd_f64(
    const double* traj, const double* mean, const double* L, int64_t L_stride,
    const int64_t* pairs, const double* tab, int M, int nbg, double tau, double delay,
    double scaled_dt, double inv_scale_g, int rel_goal, const double* times,
    int times_flags, const double* init_time, const double* init_pos,
    const double* init_vel, double reg, double* logp, double* basis_ws, int* flag_ws,
    double* work, int64_t N, int T, int P, int dof, void* stream);
int tce_pair_logprob_bwd_f32(
    const float* traj, const float* mean, const float* L, int64_t L_stride,
    const int64_t* pairs, const float* tab, int M, int nbg, float tau, float delay,
    float scaled_dt, float inv_scale_g, int rel_goal, const float* times,
    int times_flags, const float* init_time, const float* init_pos,
    const float* init_vel, float reg, const float* grad_logp, float* grad_mean,
    float* grad_L, float* basis_ws, int* flag_ws, float* work, int64_t N, int T, int P,
    int dof, void* stream);
int tce_pair_logprob_bwd_f64(
    const double* traj, const double* mean, const double* L, int64_t L_stride,
    const int64_t* pairs, const double* tab, int M, int nbg, double tau, double delay,
    double scaled_dt, double inv_scale_g, int rel_goal, const double* times,
    int times_flags, const double* init_time, const double* init_pos,
    const double* init_vel, double reg, const double* grad_logp, double* grad_mean,
    double* grad_L, double* basis_ws, int* flag_ws, double* work, int64_t N, int T,
    int P, int dof, void* stream);
/* The backward pass with the surrogate's gradient formed INSIDE the kernel:
 * grad_logp[n, p] = -exp(logp[n, p] - logp_old[n, p]) adv[n, p] / (N P)
 * (surrogate_loss, mprl/rl/agent/temporal_correlated_agent.py:641-660) from the
 * log-prob the backward kernel recomputes anyway -- neither the forward pass nor
 * tce_surrogate_* has to run before it; logp_out (nullable, [N, P]) receives that
 * log-prob, so the loss VALUE needs no forward pass either (the policy epoch
 * runs tce_surrogate_* for the record row on its second stream).  Only for the shared
 * factor's fast path known to run: L_stride 0, an affine time grid, >= 256 envs,
 * times_flags bit 3 (the caller has checked that all segments start together);
 * anything else is refused. */
int tce_pair_logprob_bwd_sur_f32(
    const float* traj, const float* mean, const float* L, int64_t L_stride,
    const int64_t* pairs, const float* tab, int M, int nbg, float tau, float delay,
    float scaled_dt, float inv_scale_g, int rel_goal, const float* times,
    int times_flags, const float* init_time, const float* init_pos,
    const float* init_vel, float reg, const float* logp_old, const float* adv,
    float* logp_out, float* grad_mean, float* grad_L, float* basis_ws, int* flag_ws,
    float* work, int64_t N, int T, int P, int dof, void* stream);
int tce_pair_logprob_bwd_sur_f64(
    const double* traj, const double* mean, const double* L, int64_t L_stride,
    const int64_t* pairs, const double* tab, int M, int nbg, double tau, double delay,
    double scaled_dt, double inv_scale_g, int rel_goal, const double* times,
    int times_flags, const double* init_time, const double* init_pos,
    const double* init_vel, double reg, const double* logp_old, const double* adv,
    double* logp_out, double* grad_mean, double* grad_L, double* basis_ws, int* flag_ws,
    double* work, int64_t N, int T, int P, int dof, void* stream);
/* Scheduling hint, process-wide: the number of compute units the caller expects
 * to be free for the kernels it is about to enqueue (0 = the whole chip, the
 * default).  The pair log-prob fast path trades latency for wave-instructions
 * only when most of the chip is free; results do not depend on it. */
int tce_set_cu_budget(int compute_units);
/* on = 0: the shared-covariance pair kernels take their general form (runtime
 * dof / basis count, per-env vectors in LDS) for every shape instead of the
 * register form built for the shipped (dof, num_basis + 1) combinations
 * (4, 6), (4, 9), (7, 9), (7, 4); default 1.  For A / B runs and tests. */
int tce_pair_env_static(int on);
/* on = 1: tce_bb_policy_epochs_f32 ends every epoch with its general finish
 * kernel also for diagonal factors (default 0: the single-round-trip kernel
 * written for them, bb_diag_finish_kernel in csrc/smlp.hip).  For A / B runs
 * and tests; results agree to rounding. */
void tce_bb_finish_general(int on);
/* on = 0: tce_policy_epoch2_* ends an epoch with the five separate launches
 * (join add, tce_chol_build_bwd, tce_adam_flat's two kernels, tce_policy_record)
 * instead of policy_tail_kernel (csrc/objective.hip; default 1, bit-identical
 * results).  For A / B runs and tests. */
void tce_policy_tail_fused(int on);
/* on = 0: the policy objective runs forward pair kernels -> tce_surrogate ->
 * backward pair kernels on the caller's stream (rounds 1 - 3); default 1: where
 * tce_pair_logprob_bwd_sur_* applies the backward forms the surrogate's gradient
 * itself and the other two, which then only feed the record row, run on the
 * objective's second stream.  Same values to rounding (the gradient's log-prob is
 * the backward kernel's recomputation). */
void tce_policy_inline_surrogate(int on);
/* out[j] = sum_n x[n, j]: gradient of a matrix shared by all envs.
 * ws: real [tce_sum_dim0_slices(N, M), M] workspace. */
int64_t tce_sum_dim0_slices(int64_t N, int64_t M);
int tce_sum_dim0_f32(const float* x, float* out, float* ws, int64_t N, int64_t M,
                     void* stream);
int tce_sum_dim0_f64(const double* x, double* out, double* ws, int64_t N, int64_t M,
                     void* stream);

/* ---- Gaussian policy head / param-space Gaussian / KL trust region -------
 * chol_build: AbstractGaussianPolicy._vector_to_cholesky
 *   (mprl/rl/policy/abstract_policy.py:166-187 -> util_matrix.py:12-33,
 *    util_numerical.py:44-68): vec [B, nvec] (nvec = K diag-only or
 *   K + K(K-1)/2) -> L [B,K,K], diag = softplus(v) + min_std, strict lower
 *   triangle filled row-major.
 * vec_env: per-env triangular-solve ops, L [.,K,K] with L_stride (0 = shared):
 *   mode 0  maha(x, y, L) = |L^-1 (x-y)|^2            (black_box_policy.py:205-224)
 *   mode 1  KL mean projection of x towards y with bound eps
 *           (trust_region_projections mean_projection; call site
 *            temporal_correlated_agent.py:530-533)
 *   mode 2  log N(x; mean = y, L L^T)                  (black_box_policy.py:95-128)
 *   bwd = 0: out ([N] or [N,K]); bwd = 1: grad_x [N,K] from grad_out (mode 2:
 *   gradient w.r.t. the mean y, plus per-env grad_L [N,K,K] when non-NULL).
 * kl_cov_part: 1/2 (tr(S_old^-1 S) - K + logdet S_old - logdet S) per matrix
 *   (gaussian_kl of trust_region_projections; temporal_correlated_agent.py:641-661).
 * kl_cov_proj: differentiable KL projection of the covariance (the job of the
 *   C++ cpp_projection / ITPAL solver, conda_env.sh:34) followed by the entropy
 *   control scaling; beta: device scalar or NULL.  ctx: double
 *   [B, tce_kl_cov_proj_ctx_len(K)] saved for the backward call.  K <= 64.
 *   warm_start != 0: ctx still holds the result of a previous call for nearby
 *   inputs (the previous policy epoch: same B, K, slightly different L): its
 *   eigenvectors start the Jacobi iteration (2-3 sweeps instead of 8-10); the
 *   first such call must find ctx zero-filled.
 */
int tce_chol_build_fwd_f32(const float* vec, float* L, int64_t B, int K, int nvec,
                           float min_std, void* stream);
int tce_chol_build_fwd_f64(const double* vec, double* L, int64_t B, int K, int nvec,
                           double min_std, void* stream);
int tce_chol_build_bwd_f32(const float* vec, const float* grad_L, float* grad_vec,
                           int64_t B, int K, int nvec, void* stream);
int tce_chol_build_bwd_f64(const double* vec, const double* grad_L, double* grad_vec,
                           int64_t B, int K, int nvec, void* stream);
int tce_vec_env_f32(int mode, int bwd, const float* x, const float* y, const float* L,
                    int64_t L_stride, float eps, const float* grad_out, float* out,
                    float* grad_x, float* grad_L, int64_t N, int K, void* stream);
int tce_vec_env_f64(int mode, int bwd, const double* x, const double* y, const double* L,
                    int64_t L_stride, double eps, const double* grad_out, double* out,
                    double* grad_x, double* grad_L, int64_t N, int K, void* stream);
/* The forward of mode 1 (mean projection) that also stores the squared
 * Mahalanobis distance it is built on, quad_out[n] = |L^-1 (x_n - y_n)|^2: the
 * policy objective's KL diagnostics take maha(new, old) and maha(proj, old) =
 * quad / s^2 from it instead of solving those two systems again.  z_out
 * (nullable; shared L only): z = L^-1 (x - y) [N, K] for
 * tce_mean_proj_bwd_acc_*, whose backward then is one substitution, not two. */
int tce_mean_proj_fwd_q_f32(const float* x, const float* y, const float* L,
                            int64_t L_stride, float eps, float* out, float* quad_out,
                            float* z_out, int64_t N, int K, void* stream);
int tce_mean_proj_fwd_q_f64(const double* x, const double* y, const double* L,
                            int64_t L_stride, double eps, double* out, double* quad_out,
                            double* z_out, int64_t N, int K, void* stream);
/* Backward of vec_env mode 2 (log N(x; y, L L^T), black_box_policy.py:95-128) for a
 * factor SHARED by all envs: grad_mean [N,K] = grad_out[n] L^-T z_n and z_out
 * [N,K] = z_n = L^-1 (x_n - y_n).  The gradient w.r.t. the shared factor is then
 * ONE product, sum_n d logp_n / d L = tril((grad_mean)^T z) - (sum_n grad_out[n])
 * diag(1 / L_ii), instead of N outer products [N,K,K] and their sum (what autograd
 * builds for log_prob under an expanded factor). */
int tce_mvn_logprob_bwd_z_f32(const float* x, const float* y, const float* L,
                              const float* grad_out, float* grad_mean, float* z_out,
                              int64_t N, int K, void* stream);
int tce_mvn_logprob_bwd_z_f64(const double* x, const double* y, const double* L,
                              const double* grad_out, double* grad_mean, double* z_out,
                              int64_t N, int K, void* stream);
/* The same with the surrogate's gradient (black_box_agent.py:468-489) formed
 * inside -- grad_out[n] = -exp(logp_n - logp_old[n]) adv[n] / N needs no sum
 * over the envs -- and the log-probs left in logp_out [N] for the loss value
 * (tce_surrogate_* with grad == NULL): policy.log_prob's forward pass and the
 * surrogate's gradient kernel drop out of the epoch. */
int tce_mvn_logprob_bwd_z_sur_f32(const float* x, const float* y, const float* L,
                                  const float* logp_old, const float* adv,
                                  float* grad_mean, float* z_out, float* logp_out,
                                  int64_t N, int K, void* stream);
int tce_mvn_logprob_bwd_z_sur_f64(const double* x, const double* y, const double* L,
                                  const double* logp_old, const double* adv,
                                  double* grad_mean, double* z_out, double* logp_out,
                                  int64_t N, int K, void* stream);
/* The backward of mode 1 (mean projection) that ADDS its result to grad_x
 * instead of storing it: the policy objective's two halves of d / d mean_new
 * (trust region loss, written on its second stream; surrogate through the
 * projection) without a separate add kernel.  The caller has waited for
 * whatever wrote grad_x. */
int tce_mean_proj_bwd_acc_f32(const float* x, const float* y, const float* L,
                              int64_t L_stride, float eps, const float* grad_out,
                              const float* z, float* grad_x, int64_t N, int K,
                              void* stream);
int tce_mean_proj_bwd_acc_f64(const double* x, const double* y, const double* L,
                              int64_t L_stride, double eps, const double* grad_out,
                              const double* z, double* grad_x, int64_t N, int K,
                              void* stream);
int tce_kl_cov_part_f32(int bwd, const float* L, const float* L_old,
                        int64_t L_old_stride, const float* grad_out, float* out,
                        float* grad_L, int64_t B, int K, void* stream);
int tce_kl_cov_part_f64(int bwd, const double* L, const double* L_old,
                        int64_t L_old_stride, const double* grad_out, double* out,
                        double* grad_L, int64_t B, int K, void* stream);
int64_t tce_kl_cov_proj_ctx_len(int K);
/* Implementation of tce_kl_cov_proj_fwd/bwd: 1 = without an
 * eigen-decomposition (csrc/klproj2.h: Newton on the dual variable with one
 * block Gauss-Jordan inversion per evaluation, products on the float64 matrix
 * instruction, implicit-function gradient in matrix form; 32 x 32 images for
 * K <= 32, 64 x 64 above); 0 = one-sided Jacobi (rounds 1 - 3); 2 (default) =
 * form 1 for K >= 20, form 0 otherwise.  Same results to ~1e-9; a context (ctx) written by
 * one form must be read by the same form; with warm_start != 0 the context must
 * come from a call with the SAME L_old. */
int tce_kl_proj_impl(int impl);
int tce_kl_cov_proj_fwd_f32(const float* L, const float* L_old, int64_t L_old_stride,
                            double eps_cov, const float* beta, int entropy_eq,
                            float* proj_L, double* ctx, int64_t B, int K, int warm_start,
                            void* stream);
int tce_kl_cov_proj_fwd_f64(const double* L, const double* L_old, int64_t L_old_stride,
                            double eps_cov, const double* beta, int entropy_eq,
                            double* proj_L, double* ctx, int64_t B, int K, int warm_start,
                            void* stream);
int tce_kl_cov_proj_bwd_f32(const float* L, const float* L_old, int64_t L_old_stride,
                            const float* proj_L, const double* ctx,
                            const float* grad_proj, float* grad_L, int64_t B, int K,
                            void* stream);
int tce_kl_cov_proj_bwd_f64(const double* L, const double* L_old, int64_t L_old_stride,
                            const double* proj_L, const double* ctx,
                            const double* grad_proj, double* grad_L, int64_t B, int K,
                            void* stream);

/* ---- rollout buffer: observation running mean/std, MDP reward ------------
 * rms_update: RunningMeanStd.update (mprl/util/util_numerical.py:315-337) of
 *   x [R, D] into the running mean/var [D] (in place; unbiased batch variance,
 *   parallel-moments merge); `count` is the running count BEFORE the update
 *   (the caller adds R afterwards, the count lives on the host like in the
 *   reference).  partials_ws: double [tce_rms_num_partials(), D, 2].
 * rms_normalize: (x - mean) / sqrt(var + eps)
 *   (mprl/rl/sampler/temporal_correlated_sampler.py:87-89), D = last dim.
 * mdp_reward: make_mdp_reward (mprl/util/util_experiment.py:261-328), in place
 *   on rewards [N,T] with the event flags [N,T] (1 byte each).
 */
int64_t tce_rms_num_partials(void);
int tce_rms_update_f32(const float* x, int64_t R, int D, float* mean, float* var,
                       double count, double* partials_ws, void* stream);
int tce_rms_update_f64(const double* x, int64_t R, int D, double* mean, double* var,
                       double count, double* partials_ws, void* stream);
int tce_rms_normalize_f32(const float* x, float* y, int64_t total, int D,
                          const float* mean, const float* var, float eps, void* stream);
int tce_rms_normalize_f64(const double* x, double* y, int64_t total, int D,
                          const double* mean, const double* var, double eps, void* stream);

/* select_batch (mprl/util/util_data_structure.py:362-375) for the black-box
 * critic's minibatches (black_box_agent.py:124-131): x_out [n, din] =
 * x[idx[i], :din] (rows of x at x_stride elements), a_out [n] = a[idx[i]],
 * b_out [n] = b[idx[i]] (b / b_out nullable); one launch. */
int tce_gather_rows_f32(const float* x, int64_t x_stride, const float* a, const float* b,
                        const int64_t* idx, int64_t n, int din, float* x_out, float* a_out,
                        float* b_out, void* stream);
int tce_gather_rows_f64(const double* x, int64_t x_stride, const double* a, const double* b,
                        const int64_t* idx, int64_t n, int din, double* x_out, double* a_out,
                        double* b_out, void* stream);
/* out [n] (int64) = a keyed pseudo-random PERMUTATION of 0 .. n-1 without a sort
 * (balanced Feistel network + cycle walking, every element on its own): the
 * critic's minibatch permutation (mprl/util/util_data_structure.py:378-391)
 * drawn on the device -- agent option minibatch_permutation: device; the default
 * is the reference's own numpy draw on the host. */
int tce_feistel_permutation(int64_t* out, int64_t n, uint64_t key, void* stream);
int tce_mdp_reward_f32(float* rewards, const uint8_t* event_flags, int64_t N, int T,
                       void* stream);
int tce_mdp_reward_f64(double* rewards, const uint8_t* event_flags, int64_t N, int T,
                       void* stream);

/* ---- exact median by radix select (csrc/select.hip) ---------------------------
 * The "median" entry of the reference's metric dictionaries, generate_stats
 * (mprl/util/util_numerical.py:130-164; per dataset tensor in
 * temporal_correlated_agent.py:166-176, black_box_agent.py:83-103), without
 * sorting the tensor: x [n] -> out[0] (double) = element of rank (n - 1) / 2 (the
 * lower median, torch.median's convention; NaNs order above +inf).  ws: unsigned
 * [tce_median_ws_len()], zeroed once by the caller (each call leaves it zeroed);
 * one workspace per stream.  1 <= n < 2^32.
 */
int tce_median_ws_len(void);
int tce_median_f32(const float* x, int64_t n, double* out, unsigned* ws, void* stream);
int tce_median_f64(const double* x, int64_t n, double* out, unsigned* ws, void* stream);
/* All five entries of generate_stats (mprl/util/util_numerical.py:130-164) for
 * one tensor in one chain of launches: out5 (double) = {mean, max, min, median
 * (as tce_median_*), standard deviation with n - 1} of x [n].  The first pass of
 * the select reads every element anyway and carries the sums / extrema (double
 * accumulation of x - x[0], per-workgroup partials combined in a fixed order).
 * ws: unsigned [tce_stats5_ws_len()], 8-byte aligned, zeroed once by the caller;
 * one workspace per stream. */
int tce_stats5_ws_len(void);
int tce_stats5_f32(const float* x, int64_t n, double* out5, unsigned* ws, void* stream);
int tce_stats5_f64(const double* x, int64_t n, double* out5, unsigned* ws, void* stream);

/* ---- GPU-resident synthetic env suite ------------------------
 * Stands in for the env processes behind SubprocVecEnv.step
 * (mprl/util/util_mp.py:119-185) and speaks the per-episode protocol the
 * sampler consumes (mprl/rl/sampler/temporal_correlated_sampler.py:226-303):
 * ONE launch steps N envs through a whole episode of T steps.
 *   actions  [N,T,2 dof]  desired pos | vel (the ProDMP trajectory)
 *   init_obs [N,D]        reset observation, D = d_task + 1 + 2 dof, laid out
 *                         [q(dof) | qd(dof) | obj(3) | goal(3) | 0.. | time |
 *                          des_pos(dof) | des_vel(dof)]
 *   family   0 reach, 1 push / box-push, 2 table-tennis-like (event hit_ball),
 *            3 hopper-jump-like (event has_left_floor)
 * Dynamics: unit point mass per dof under PD tracking, semi-implicit Euler:
 *   a = kp (des_pos - q) + kd (des_vel - qd); qd += dt a; q += dt qd.
 * Outputs: states [N,T+1,D] (row 0 = init_obs; NULL = not wanted, the
 * black-box env), rewards [N,T], event_flags [N,T] (1 byte; may be NULL),
 * metrics [N,2] = {success, final distance} (may be NULL), and -- when
 * moment_partials (double [N,D,2]) is given -- per-env shifted column sums
 * {sum(x - shift), sum((x - shift)^2)} over the T+1 rows, which
 * tce_rms_merge_* folds into the running mean/var exactly like
 * tce_rms_update_* (RunningMeanStd.update, mprl/util/util_numerical.py:315-337;
 * `shift` may alias `mean`, batch_count = number of rows summed).
 */
int tce_env_rollout_f32(const float* actions, const float* init_obs, int family,
                        int64_t N, int T, int dof, int d_task, float dt, float kp,
                        float kd, float* states, float* rewards,
                        uint8_t* event_flags, float* metrics, const float* shift,
                        double* moment_partials, void* stream);
int tce_env_rollout_f64(const double* actions, const double* init_obs, int family,
                        int64_t N, int T, int dof, int d_task, double dt, double kp,
                        double kd, double* states, double* rewards,
                        uint8_t* event_flags, double* metrics, const double* shift,
                        double* moment_partials, void* stream);
int tce_rms_merge_f32(const double* moment_partials, int64_t nparts, int D,
                      const float* shift, double batch_count, double count,
                      float* mean, float* var, void* stream);
int tce_rms_merge_f64(const double* moment_partials, int64_t nparts, int D,
                      const double* shift, double batch_count, double count,
                      double* mean, double* var, void* stream);

/* ---- streams restricted to a slice of every XCD ---------------------------
 * The critic update and the policy update of one iteration are independent
 * (mprl/rl/agent/temporal_correlated_agent.py:55-70 runs them back to back);
 * here they run side by side on disjoint compute units.  The stream's kernels
 * use units [first_cu, first_cu + cus_per_xcd) of each of the 8 XCDs.
 */
int tce_stream_create_cu_range(int first_cu, int cus_per_xcd, void** stream);
/* No reference counterpart (measurement aid): an empty launch of `tag` workgroups
 * named tce_marker_kernel -- a mark that a rocprofv3 kernel trace shows, so that
 * a summary can be restricted to the dispatches between two marks (bench.py puts
 * tag 1 / 2 around its timed steps; scripts/rocpd_stats.py --between-markers). */
int tce_marker(int tag, void* stream);
/* No reference counterpart (scheduling aid): one wave that spins for `us`
 * microseconds of wall clock on `stream` -- the probe with which
 * tce_rl_amd/streams.py finds out whether two streams run side by side (they do
 * not when they were bound to the same hardware queue). 0 < us <= 100000. */
int tce_spin_us(double us, void* stream);
int tce_stream_destroy(void* stream);

/* ---- policy objective, shared (non-contextual) covariance ------------------
 * surrogate: out[0] = -mean(exp(lp_new - lp_old) * adv), out[1] = mean ratio
 *   (surrogate_loss, mprl/rl/agent/temporal_correlated_agent.py:718-739);
 *   grad_lp (nullable) [M] = d out[0] / d lp_new.  ws: double
 *   [tce_surrogate_ws_len()], ws[0] ZEROED ONCE by the caller before the first
 *   call (block ticket, re-armed by every call), reusable across calls on one
 *   stream.
 * kl_shared: one call for what update_policy evaluates per epoch besides the
 *   surrogate (temporal_correlated_agent.py:530-567,641-686,741-745):
 *   out16[0..11] = means over envs of gaussian_kl_details (mean, cov, shape,
 *   volume parts) for the pairs (new || old), (new || proj), (proj || old);
 *   out16[12] = entropy of the projected policy; out16[13] = trust region loss
 *   tr_coeff * mean(KL_mean(new || proj) [+ KL_cov(new || proj) if
 *   tr_include_cov]) (projection layer get_trust_region_loss, third-party);
 *   grad_mean [N,K] / grad_L [K,K] (nullable) = its gradients w.r.t. mean_new /
 *   L_new (proj treated as constant).  Means [N,K]; L_* ONE lower-triangular
 *   [K,K] factor each; ws: double [tce_kl_shared_ws_len(N)].
 */
int64_t tce_kl_shared_ws_len(int64_t N);
int64_t tce_surrogate_ws_len(void);
int tce_surrogate_f32(const float* lp_new, const float* lp_old, const float* adv, int64_t M,
                      float* out, float* grad_lp, double* ws, void* stream);
int tce_surrogate_f64(const double* lp_new, const double* lp_old, const double* adv, int64_t M,
                      double* out, double* grad_lp, double* ws, void* stream);
int tce_kl_shared_f32(const float* mean_new, const float* mean_old, const float* mean_proj,
                      const float* L_new, const float* L_old, const float* L_proj, int64_t N,
                      int K, float tr_coeff, int tr_include_cov, float* out16,
                      float* grad_mean, float* grad_L, double* ws, void* stream);
int tce_kl_shared_f64(const double* mean_new, const double* mean_old, const double* mean_proj,
                      const double* L_new, const double* L_old, const double* L_proj, int64_t N,
                      int K, double tr_coeff, int tr_include_cov, double* out16,
                      double* grad_mean, double* grad_L, double* ws, void* stream);

/* The K x K half of tce_kl_shared alone: the caller supplies the sums over
 * the envs of the three squared Mahalanobis terms |L_old^-1 (new - old)|^2,
 * |L_proj^-1 (new - proj)|^2, |L_old^-1 (proj - old)|^2 (partials: double
 * [nparts][3], added in order). */
int tce_kl_shared_mat_f32(const float* L_new, const float* L_old, const float* L_proj,
                          int64_t N, int K, float tr_coeff, int tr_include_cov,
                          const double* partials, int nparts, float* out16, float* grad_L,
                          void* stream);
int tce_kl_shared_mat_f64(const double* L_new, const double* L_old, const double* L_proj,
                          int64_t N, int K, double tr_coeff, int tr_include_cov,
                          const double* partials, int nparts, double* out16, double* grad_L,
                          void* stream);

/* The whole objective of one TCE policy epoch and its gradient in ONE call
 * (shared covariance, KL projection; mprl/rl/agent/temporal_correlated_agent.py:
 * 523-612): mean projection -> covariance projection -> pair log-prob ->
 * surrogate -> their backward passes -> KL diagnostics / entropy / trust
 * region loss, i.e. tce_vec_env(mode 1), tce_kl_cov_proj_fwd, tce_pair_logprob_fwd,
 * tce_surrogate, tce_pair_logprob_bwd, tce_kl_shared, tce_vec_env bwd,
 * tce_kl_cov_proj_bwd with the arguments those take.  The single-workgroup
 * K x K kernels run on a library-owned second stream beside the per-env kernels
 * (fork / join by events on `stream`; works under stream capture).
 * Outputs: grad_mean [N,K], grad_L [K,K] = d(surrogate + trust region loss
 * - ent_coef * entropy) / d(mean_new, L_new); sur2 [2] as tce_surrogate; out16
 * as tce_kl_shared.  ws: scratch of tce_policy_objective_ws_len(N, K, P)
 * elements; pair_work / basis_ws / flag_ws / proj_ctx / sur_ws / kl_ws as in
 * the separate calls (times_flags_fwd / _bwd: the times_flags of the two pair
 * log-prob calls).  proj_started != 0: see tce_policy_objective_begin_*;
 * defer_join & 1: see tce_policy_objective_end_*; defer_join & 2 ("split", for
 * the balance check of temporal_correlated_agent.py:447-522): nothing is added
 * up -- grad_mean / grad_L receive the trust region loss's gradient alone, the
 * surrogate's stays in ws (d / d mean_new at ws + 2 up4(N K); d / d L_new at
 * ws + 3 up4(N K) + 2 up4(N P) + 2 up4(K K), complete once the library's side
 * stream has been joined; the entropy term's d / d L_new one up4(K K) block
 * behind it), up4 = rounded up to a multiple of 4. */
int64_t tce_policy_objective_ws_len(int64_t N, int K, int P);
/* n = 1: everything on the caller's stream (no second stream, no events); n = 2
 * (default): as described above.  A process that already drives more streams
 * than the device has hardware queues (e.g. the critic stream, the policy
 * stream and the streams of two RCCL communicators) should say 1: streams that
 * share a hardware queue wait for each other's kernels. */
int tce_policy_objective_streams(int n);
/* The library-owned second stream (created on first use): lets a caller order
 * its own work against it -- the tests stall it to show that the deferred join
 * of tce_policy_objective_* does not depend on timing. */
int tce_policy_objective_side_stream(void** stream);
/* Hand the library the stream to use as its second one (instead of creating
 * its own on first use): a HIP stream is bound to a hardware queue when it is
 * created and streams that share a queue wait for each other, so the caller
 * picks one it has PROBED to run beside its other streams (tce_spin_us below;
 * tce_rl_amd/streams.py).  The stream must outlive the library's use of it. */
int tce_policy_objective_use_stream(void* stream);
int tce_policy_objective_f32(
    const float* mean_new, const float* L_new, const float* mean_old, const float* L_old,
    const float* traj, const float* logp_old, const float* adv, const int64_t* pairs,
    const float* tab, int M, int nbg, float tau, float delay, float scaled_dt,
    float inv_scale_g, int rel_goal, const float* times, int times_flags_fwd,
    int times_flags_bwd, const float* init_time, const float* init_pos,
    const float* init_vel, float reg, float* basis_ws, int* flag_ws, float* pair_work,
    float eps_mean, double eps_cov, const float* beta, int entropy_eq, double* proj_ctx,
    float tr_coeff, int tr_include_cov, float ent_coef, double* sur_ws, double* kl_ws,
    float* ws, float* grad_mean, float* grad_L, float* sur2, float* out16, int64_t N, int T,
    int P, int dof, int K, int proj_started, int defer_join, void* stream);
int tce_policy_objective_f64(
    const double* mean_new, const double* L_new, const double* mean_old, const double* L_old,
    const double* traj, const double* logp_old, const double* adv, const int64_t* pairs,
    const double* tab, int M, int nbg, double tau, double delay, double scaled_dt,
    double inv_scale_g, int rel_goal, const double* times, int times_flags_fwd,
    int times_flags_bwd, const double* init_time, const double* init_pos,
    const double* init_vel, double reg, double* basis_ws, int* flag_ws, double* pair_work,
    double eps_mean, double eps_cov, const double* beta, int entropy_eq, double* proj_ctx,
    double tr_coeff, int tr_include_cov, double ent_coef, double* sur_ws, double* kl_ws,
    double* ws, double* grad_mean, double* grad_L, double* sur2, double* out16, int64_t N,
    int T, int P, int dof, int K, int proj_started, int defer_join, void* stream);
/* Optional first half: the Cholesky head (tce_chol_build_fwd of var_vec -> L_new
 * [K,K]) and the covariance projection of the coming tce_policy_objective call
 * (same ws / proj_ctx / N / K / P, proj_started = 1, L_new = this output) are
 * put on the second stream at once -- they depend on the variance parameters
 * only and run beside whatever `stream` does until that call (the forward pass
 * of the mean net). */
int tce_policy_objective_begin_f32(const float* var_vec, int nvec, float min_std,
                                   const float* L_old, double eps_cov, const float* beta,
                                   int entropy_eq, double* proj_ctx, float* L_new, float* ws,
                                   int64_t N, int K, int P, void* stream);
int tce_policy_objective_begin_f64(const double* var_vec, int nvec, double min_std,
                                   const double* L_old, double eps_cov, const double* beta,
                                   int entropy_eq, double* proj_ctx, double* L_new,
                                   double* ws, int64_t N, int K, int P, void* stream);
/* Optional last half (defer_join != 0 in tce_policy_objective): that call then
 * returns with grad_mean complete but the side stream still working on the
 * backward pass of the covariance projection -- grad_L holds the trust-region
 * part only -- so that `stream` can go on with the mean net's backward pass;
 * this call joins the side stream and adds the missing part to grad_L.  It must
 * follow every deferred call (before the next objective call, and before the
 * end of a stream capture). */
int tce_policy_objective_end_f32(float* grad_L, float* ws, int64_t N, int K, int P,
                                 void* stream);
int tce_policy_objective_end_f64(double* grad_L, double* ws, int64_t N, int K, int P,
                                 void* stream);

/* The same for the black-box agent (mprl/rl/agent/black_box_agent.py:283-357):
 * the log-prob is BlackBoxPolicy.log_prob of the sampled parameter vectors
 * `actions` [N,K] under the projected Gaussian (tce_vec_env mode 2) instead of
 * the pair-wise trajectory log-prob; logp_old / adv [N].  proj_mean_out [N,K] /
 * proj_L_out [K,K] (nullable) receive the projected distribution.  ws: scratch
 * of tce_bb_policy_objective_ws_len(N, K) elements. */
int64_t tce_bb_policy_objective_ws_len(int64_t N, int K);
int tce_bb_policy_objective_f32(
    const float* mean_new, const float* L_new, const float* mean_old, const float* L_old,
    const float* actions, const float* logp_old, const float* adv, float eps_mean,
    double eps_cov, const float* beta, int entropy_eq, double* proj_ctx, float tr_coeff,
    int tr_include_cov, float ent_coef, double* sur_ws, double* kl_ws, float* ws,
    float* grad_mean, float* grad_L, float* sur2, float* out16, float* proj_mean_out,
    float* proj_L_out, int64_t N, int K, void* stream);
int tce_bb_policy_objective_f64(
    const double* mean_new, const double* L_new, const double* mean_old, const double* L_old,
    const double* actions, const double* logp_old, const double* adv, double eps_mean,
    double eps_cov, const double* beta, int entropy_eq, double* proj_ctx, double tr_coeff,
    int tr_include_cov, double ent_coef, double* sur_ws, double* kl_ws, double* ws,
    double* grad_mean, double* grad_L, double* sur2, double* out16, double* proj_mean_out,
    double* proj_L_out, int64_t N, int K, void* stream);
/* Gradient of a final Linear layer y = h W^T + b (the mean net's output layer,
 * mprl/util/util_nn.py:225-246 under autograd) from grad_out = dL/dy [N,K] and
 * hidden = h [N,H]: grad_W [K,H] = grad_out^T h, grad_b [K] = sum_n grad_out.
 * K <= 64, H <= 256; ws: real [tce_out_layer_grad_ws_len(N, K, H)].
 * policy_record: the per-epoch record row {surrogate, entropy loss, trust
 * region loss, total, entropy, |g|, |g| clipped, 12 KL means} (the quantities
 * temporal_correlated_agent.py:598-636 appends per epoch) from the outputs of
 * tce_policy_objective and the optimizer's two gradient norms, without a host
 * round trip. */
int64_t tce_out_layer_grad_ws_len(int64_t N, int K, int H);
int tce_out_layer_grad_f32(const float* grad_out, const float* hidden, float* grad_W,
                           float* grad_b, float* ws, int64_t N, int K, int H, void* stream);
int tce_out_layer_grad_f64(const double* grad_out, const double* hidden, double* grad_W,
                           double* grad_b, double* ws, int64_t N, int K, int H, void* stream);
int tce_policy_record_f32(const float* sur2, const float* out16, const float* norms2,
                          float ent_coef, float* row19, void* stream);
int tce_policy_record_f64(const double* sur2, const double* out16, const double* norms2,
                          double ent_coef, double* row19, void* stream);

/* One whole TCE policy epoch WITHOUT autograd in one call (the loop body of
 * update_policy, mprl/rl/agent/temporal_correlated_agent.py:523-612, for a
 * non-contextual covariance and a D_in <= 40 -> 128 -> 128 -> K float32 mean
 * net): Cholesky head + covariance projection (second stream), mean net
 * forward (tce_mlp_hidden_f32 + tce_lin_rows_f32), tce_policy_objective_f32
 * (deferred join), the parameter gradients written straight into the flat
 * gradient `grad` (tce_lin_rows_f32, tce_mlp_hidden_f32 backward,
 * tce_out_layer_grad, tce_policy_objective_end, tce_chol_build_bwd), then --
 * do_adam -- tce_adam_flat on all parameters and tce_policy_record into
 * rec_row19.  param / grad / m / v: FLAT in the order W1 [128][din] | b1 | W2 |
 * b2 | W3 [K][128] | b3 | variance vector [nvec].  x [N, din] rows with stride
 * x_stride.  The objective's arguments (mean_old ... obj_ws) are those of
 * tce_policy_objective_f32 (obj_ws = its `ws`); ws: float
 * [tce_policy_epoch_ws_len(N, K)]; partials: float [min(tce_mlp_critic_grid(),
 * ceil(N / 64))][tce_mlp_critic_num_params(din) + 2]; ol_ws: float
 * [tce_out_layer_grad_ws_len(N, K, 128)].  After the call ws holds, behind
 * 2 N 128 floats, mean_new [N,K] and its gradient.  do_adam == 0: the caller
 * (env shards) all-reduces `grad`, steps and records itself. */
int64_t tce_policy_epoch_ws_len(int64_t N, int K);
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
    void* stream);

/* The same epoch for every mean net the library has kernels for, float32 and
 * float64, and for the epochs of a balance-check iteration.
 * net_kind 0: the float32 D_in <= 40 -> 128 -> 128 -> K path above (hidden = 128,
 * num_hidden = 2; partials / ol_ws as above); net_kind 1: the row kernels of
 * csrc/pmlp.hip (tce_pmlp_supported(din, hidden, num_hidden, K, element size):
 * box pushing's float64 128 x 2 net,
 * mprl/config/box_push_random_init/tcp/entire/shared.yaml:7,75-78, table tennis's
 * 256 x 1 tanh net, mprl/config/table_tennis_4d/tcp/entire/shared.yaml:78-81;
 * partials: [tce_pmlp_max_slabs()][tce_pmlp_num_params(...)], ol_ws unused).
 * param / grad / m / v: FLAT, the mean net's parameters in MLP.parameters()
 * order, then the variance vector [nvec]; nparam = their total count.
 * ws: [tce_policy_epoch2_ws_len(N, K, hidden, nparam)].
 * balance != 0: the epoch of an iteration with num_iterations % balance_check
 * == 1 (mprl/rl/agent/temporal_correlated_agent.py:447-522: there two extra
 * forward / backward passes, one of the surrogate loss alone, one of the trust
 * region loss alone, each followed by grad_norm_clip(0, params) for its norm).
 * Here the objective is evaluated once with its gradient kept in two parts
 * (tce_policy_objective_* with defer_join = 3); each part goes back through the
 * mean net and the Cholesky head alone, bal2[0] / bal2[1] receive the norms of
 * the surrogate's / the trust region loss's parameter gradient, and their sum
 * (plus the entropy term's) is the gradient the optimizer step uses.  Requires
 * do_adam.
 * xchg (nullable; needs do_adam): the envs are sharded over the ranks of this
 * exchange (tce_xchg_create) -- the kernel that ends the epoch adds the peers'
 * gradients in rank order before Adam (grad_scale = 1 / world), so a sharded
 * epoch is the same ONE call; the two norms of a balance epoch are those of the
 * rank-averaged parts (two more collectives).  grad keeps the SUMMED gradient. */
int64_t tce_policy_epoch2_ws_len(int64_t N, int K, int hidden, int64_t nparam);
int tce_policy_epoch2_f32(
    const float* x, int64_t x_stride, int64_t N, int din, int hidden, int num_hidden,
    int net_kind, int act, int nvec, float min_std, float* param, float* grad,
    const float* mean_old, const float* L_old, const float* traj, const float* logp_old,
    const float* adv, const int64_t* pairs, const float* tab, int M, int nbg, float tau,
    float delay, float scaled_dt, float inv_scale_g, int rel_goal, const float* times,
    int times_flags_fwd, int times_flags_bwd, const float* init_time, const float* init_pos,
    const float* init_vel, float reg, float* basis_ws, int* flag_ws, float* pair_work,
    float eps_mean, double eps_cov, const float* beta, int entropy_eq, double* proj_ctx,
    float tr_coeff, int tr_include_cov, float ent_coef, double* sur_ws, double* kl_ws,
    float* obj_ws, float* ws, float* partials, float* ol_ws, int T, int P, int dof, int K,
    float* m, float* v, float* opt_state, float lr, float beta1, float beta2, float eps,
    float weight_decay, float clip_grad, float grad_scale, int do_adam, int balance,
    float* rec_row19, float* bal2, void* xchg, void* stream);
int tce_policy_epoch2_f64(
    const double* x, int64_t x_stride, int64_t N, int din, int hidden, int num_hidden,
    int net_kind, int act, int nvec, double min_std, double* param, double* grad,
    const double* mean_old, const double* L_old, const double* traj, const double* logp_old,
    const double* adv, const int64_t* pairs, const double* tab, int M, int nbg, double tau,
    double delay, double scaled_dt, double inv_scale_g, int rel_goal, const double* times,
    int times_flags_fwd, int times_flags_bwd, const double* init_time, const double* init_pos,
    const double* init_vel, double reg, double* basis_ws, int* flag_ws, double* pair_work,
    double eps_mean, double eps_cov, const double* beta, int entropy_eq, double* proj_ctx,
    double tr_coeff, int tr_include_cov, double ent_coef, double* sur_ws, double* kl_ws,
    double* obj_ws, double* ws, double* partials, double* ol_ws, int T, int P, int dof, int K,
    double* m, double* v, double* opt_state, double lr, double beta1, double beta2, double eps,
    double weight_decay, double clip_grad, double grad_scale, int do_adam, int balance,
    double* rec_row19, double* bal2, void* xchg, void* stream);

/* One whole policy epoch of the black-box agent in one call, for the mean nets
 * the 64-wide row kernels (tce_bb_policy_epochs_f32) do not cover:
 * mprl/rl/agent/black_box_agent.py:225-339 with the nets of
 * mprl/config/box_push_random_init/bbrl/entire/shared.yaml:66-67 (128 x 2) and
 * mprl/config/table_tennis_4d/bbrl/entire/shared.yaml:72-73 (256 x 1).
 * Replaces per epoch: policy.policy(states), projection(...), policy.log_prob,
 * surrogate_loss, kl_old_new_proj, entropy_loss, get_trust_region_loss,
 * zero_grad / backward, grad_norm_clip, optimizer.step -- and, with balance != 0
 * (:226-284), the two extra passes whose gradient norms go to bal2.
 * net_kind / hidden / num_hidden / param / grad / partials / ol_ws / ws / rec_row19 /
 * balance / do_adam / xchg: as tce_policy_epoch2_*; obj_ws:
 * [tce_bb_policy_objective_ws_len(N, K)]; the other objective arguments as
 * tce_bb_policy_objective_*.  After the call ws holds this epoch's mean_new [N,K]
 * at 2 up4(N hidden) and L_new [K,K] at 2 up4(N hidden) + 2 up4(N K) (up4: rounded
 * up to a multiple of 4); proj_mean_out [N,K] / proj_L_out [K,K] (nullable)
 * receive the projected distribution. */
int tce_bb_policy_epoch_f32(
    const float* x, int64_t x_stride, int64_t N, int din, int hidden, int num_hidden,
    int net_kind, int act, int nvec, float min_std, float* param, float* grad,
    const float* mean_old, const float* L_old, const float* actions, const float* logp_old,
    const float* adv, float eps_mean, double eps_cov, const float* beta, int entropy_eq,
    double* proj_ctx, float tr_coeff, int tr_include_cov, float ent_coef, double* sur_ws,
    double* kl_ws, float* obj_ws, float* ws, float* partials, float* ol_ws, int K, float* m,
    float* v, float* opt_state, float lr, float beta1, float beta2, float eps,
    float weight_decay, float clip_grad, float grad_scale, int do_adam, int balance,
    float* rec_row19, float* bal2, float* proj_mean_out, float* proj_L_out, void* xchg,
    void* stream);
int tce_bb_policy_epoch_f64(
    const double* x, int64_t x_stride, int64_t N, int din, int hidden, int num_hidden,
    int net_kind, int act, int nvec, double min_std, double* param, double* grad,
    const double* mean_old, const double* L_old, const double* actions, const double* logp_old,
    const double* adv, double eps_mean, double eps_cov, const double* beta, int entropy_eq,
    double* proj_ctx, double tr_coeff, int tr_include_cov, double ent_coef, double* sur_ws,
    double* kl_ws, double* obj_ws, double* ws, double* partials, double* ol_ws, int K, double* m,
    double* v, double* opt_state, double lr, double beta1, double beta2, double eps,
    double weight_decay, double clip_grad, double grad_scale, int do_adam, int balance,
    double* rec_row19, double* bal2, double* proj_mean_out, double* proj_L_out, void* xchg,
    void* stream);

/* The two hidden layers D_in -> 128 -> 128 (fp32) of a network with a wider
 * output -- the policy mean net (mprl/rl/policy/abstract_policy.py:58-99 ->
 * mprl/util/util_nn.py:225-246) -- on the kernels of the fused critic epoch.
 * grad_hidden == NULL: forward, hidden_out [R][128].  Otherwise backward of
 * dL/dH2 = grad_hidden [R][128] (forward recomputed): grad [tce_mlp_critic_
 * num_params(din)] receives dW1, db1, dW2, db2 in that order (trailing w3 / b3
 * slots zero); partials as above; stats float[2] scratch zeroed by the caller. */
int tce_mlp_hidden_f32(const float* x, int64_t env_stride, int64_t row_stride, int T,
                       int64_t R, int din, const float* w1, const float* b1,
                       const float* w2, const float* b2, int act, const float* grad_hidden,
                       float* hidden_out, float* partials, float* grad, float* stats,
                       void* stream);

/* ---- optimizer: flat Adam with global-norm clipping -----------------------
 * One optimizer step over a flat parameter buffer of n elements:
 *   grad_norm_clip(clip, params)   mprl/util/util_numerical.py:244-275
 *     (coef = min(1, clip / (|g| + 1e-6)) as torch.nn.utils.clip_grad_norm_;
 *      clip <= 0: norms only)
 *   torch.optim.Adam(lr, betas, eps, weight_decay).step()   -- L2-in-gradient
 *     weight decay, mprl/rl/agent/abstract_agent.py:62-82; used at
 *     temporal_correlated_agent.py:361-366,597-612, black_box_agent.py:160-166,330-339.
 * state: 4 elements on the device = {step count (incremented by the call),
 *   |g| before clipping, |g| after, clip factor}.  sumsq_in (nullable): |g|^2
 *   already reduced by the producer (tce_mlp_critic_f32 stats[1]).  grad_scale:
 *   the gradient is taken as grad_scale * grad (1 / world size after a summing
 *   all-reduce of the env shards' gradients), norms included.
 */
int tce_adam_flat_f32(float* param, const float* grad, float* m, float* v, int64_t n,
                      float* state, const float* sumsq_in, float lr, float beta1, float beta2,
                      float eps, float weight_decay, float clip, float grad_scale,
                      void* stream);
int tce_adam_flat_f64(double* param, const double* grad, double* m, double* v, int64_t n,
                      double* state, const double* sumsq_in, double lr, double beta1,
                      double beta2, double eps, double weight_decay, double clip,
                      double grad_scale, void* stream);

/* The same step (grad_norm_clip + Adam) as ONE launch for buffers of at most
 * 2^17 elements -- what a sharded update runs behind each of its ~100 gradient
 * all-reduces per iteration.  step: the step count INCLUDING this update, kept
 * by the caller (stored to state[0]); norms_out (nullable) [2] receives {|g|
 * before, |g| after clipping} (also in state[1:3]). */
int tce_adam_once_f32(float* param, const float* grad, float* m, float* v, int64_t n,
                      float* state, float* norms_out, float step, float lr, float beta1,
                      float beta2, float eps, float weight_decay, float clip, float grad_scale,
                      void* stream);
int tce_adam_once_f64(double* param, const double* grad, double* m, double* v, int64_t n,
                      double* state, double* norms_out, double step, double lr, double beta1,
                      double beta2, double eps, double weight_decay, double clip,
                      double grad_scale, void* stream);

/* ---- one-shot gradient exchange between the env shards of one node -------
 * No reference counterpart: mprl/ is single-device (no collectives anywhere);
 * the sum over ranks stands for the global-batch `.mean()` of its losses
 * (mprl/rl/agent/temporal_correlated_agent.py:716,736;
 * mprl/rl/agent/black_box_agent.py:160-166,330-339) when the envs of one job
 * are sharded over the GPUs of a node.  Replaces, in the sharded
 * update, torch.distributed.all_reduce between two C calls: every rank owns
 * one peer-visible buffer (uncached device memory, mapped by the peers through
 * HIP IPC over xGMI); the kernels that finish an epoch publish their locally
 * reduced gradient there, wait for the peers' per-workgroup flags, add the
 * peers' values in rank order (bit-identical sums on every rank) and apply
 * Adam -- no separate collective launch, no host round trip (csrc/xchg.h).
 *   tce_xchg_create: this rank's end; max_bytes = largest message (bytes of
 *     the largest flat gradient).  world <= 8.  *out receives the handle that
 *     every `void* xchg` argument below takes (NULL there = no exchange).
 *   tce_xchg_export / tce_xchg_connect: the 64-byte (tce_xchg_handle_bytes)
 *     IPC handle of the own buffer; the handles of all ranks in rank order
 *     (exchanged by the caller over any channel, e.g. an all-gather) map the
 *     peers.  tce_xchg_connect_local: a peer that lives in the same process.
 *   tce_xchg_status: 0, or 1 + r once a wait for rank r exceeded the limit
 *     (TCE_XCHG_TIMEOUT_MS, default 20000; tce_xchg_set_timeout_ms): the
 *     kernel goes on instead of hanging, the caller must treat it as fatal.
 *   tce_xchg_counters: collectives issued and payload bytes, per rank.
 *   tce_xchg_wait_stats: how long workgroup 0 of this rank's collectives waited
 *     for its slowest peer since the last reset (sum / maximum in microseconds,
 *     number of collectives): what a straggler costs, per exchange.  Blocking
 *     (a device -> host copy of four words); reset != 0 clears the words.
 *   tce_xchg_allreduce_*: in-place sum over ranks of buf [n], rank order.
 *   tce_xchg_allgather_f64: out [world][n] = every rank's mine [n] (one
 *     workgroup; the small per-step statistics of a sharded run).
 *   tce_xchg_adam_*: all-reduce of grad (the sum stays there) + tce_adam_once_*
 *     in ONE launch (clip == 0; with clipping: all-reduce, then the step).
 * Every rank must issue the same sequence of collectives on an exchange, from
 * one stream; two independent chains (critic / policy epochs) take two
 * exchanges.  A launch that carries a sequence number cannot be replayed from
 * a HIP graph.
 */
int tce_xchg_handle_bytes(void);
int tce_xchg_create(int rank, int world, int64_t max_bytes, void** out);
int tce_xchg_export(void* xchg, void* handle_out);
int tce_xchg_connect(void* xchg, const void* handles);
int tce_xchg_connect_local(void* xchg, int peer_rank, void* peer_xchg);
int tce_xchg_destroy(void* xchg);
int tce_xchg_status(void* xchg);
int tce_xchg_set_timeout_ms(void* xchg, double ms);
int tce_xchg_counters(void* xchg, int64_t* collectives, int64_t* bytes);
int tce_xchg_wait_stats(void* xchg, double* total_us, double* max_us, int64_t* collectives, int reset);
int tce_xchg_allgather_f64(void* xchg, const double* mine, double* out, int64_t n, void* stream);
int tce_xchg_allreduce_f32(void* xchg, float* buf, int64_t n, void* stream);
int tce_xchg_allreduce_f64(void* xchg, double* buf, int64_t n, void* stream);
/* the same out of place: dst [n] = sum over ranks of src [n] (src is left as it is) */
int tce_xchg_allreduce_to_f32(void* xchg, const float* src, float* dst, int64_t n, void* stream);
int tce_xchg_allreduce_to_f64(void* xchg, const double* src, double* dst, int64_t n,
                              void* stream);
int tce_xchg_adam_f32(void* xchg, float* param, float* grad, float* m, float* v, int64_t n,
                      float* state, float* norms_out, float step, float lr, float beta1,
                      float beta2, float eps, float weight_decay, float clip, float grad_scale,
                      void* stream);
int tce_xchg_adam_f64(void* xchg, double* param, double* grad, double* m, double* v, int64_t n,
                      double* state, double* norms_out, double step, double lr, double beta1,
                      double beta2, double eps, double weight_decay, double clip,
                      double grad_scale, void* stream);

/* ---- fused critic MLP epoch (exact-fp32 MFMA) ----------------------------
 * Forward (+ value loss + backward when `partials` != NULL) of the value
 * network D_in -> 128 -> 128 -> 1 (ValueFunction.critic ->
 * mprl/util/util_nn.py:225-246; value_loss + backward of one critic epoch,
 * mprl/rl/agent/temporal_correlated_agent.py:343-366,688-716) over R rows.
 * Row r = (n, t) lives at x + n*env_stride + t*row_stride (t < T, the first
 * D_in <= 40 features are used), so the critic reads the rollout buffer in
 * place.  Weights in torch.nn.Linear layout [out][in].  act: 0 tanh, 1 relu,
 * 2 leaky_relu(0.01), 3 softplus.  clip > 0: PPO-style clipped value loss with
 * old_values.  values (nullable) [R]; partials float
 * [tce_mlp_critic_grid(), num_params + 2]; grad float [num_params] in the order
 * W1, b1, W2, b2, w3, b3; stats float[2] = {mean loss, |grad|^2}, zeroed by the
 * caller ([1] accumulates).  adam_param != NULL: the Adam step of the critic
 * optimizer (torch.optim.Adam with L2 weight decay, abstract_agent.py:62-82;
 * temporal_correlated_agent.py:361-366) on the flat buffers adam_param / adam_m /
 * adam_v [num_params] is fused into the gradient reduction (no clipping;
 * adam_step = step count including this update, stored to adam_state[0]).
 * xchg (nullable; needs adam_param): the envs are sharded over the ranks of
 * this exchange (tce_xchg_*) -- the slab reduction leaves the local gradient and
 * ONE small launch (tce_xchg_adam_*: few workgroups wait for the peers, not the
 * reduction's hundreds) adds the peers' gradients in rank order and applies Adam
 * with grad_scale (1 / world): still one call per epoch.  stats is then float[4]:
 * [2], [3] receive |g| of the rank-averaged gradient (before / after clipping);
 * grad keeps the SUMMED gradient; stats[0] / [1] stay the local shard's.
 * max_workgroups (0 = one per CU): persistent workgroups to launch; fewer than
 * the CU count leaves CUs free for kernels of another stream (the policy
 * update runs beside the critic epochs).
 */
int tce_mlp_critic_hidden(void);
int tce_mlp_critic_grid(void);
int64_t tce_mlp_critic_num_params(int din);
int tce_mlp_critic_f32(const float* x, int64_t env_stride, int64_t row_stride, int T,
                       int64_t R, int din, const float* w1, const float* b1,
                       const float* w2, const float* b2, const float* w3, const float* b3,
                       int act, const float* returns, const float* old_values, float clip,
                       float* values, float* partials, float* grad, float* stats,
                       int max_workgroups, float* adam_param, float* adam_m, float* adam_v,
                       float* adam_state, float lr, float beta1, float beta2, float eps,
                       float weight_decay, float adam_step, float grad_scale, void* xchg,
                       void* stream);

/* One critic EPOCH in minibatches -- the reference's class default is
 * num_minibatchs = 10 (mprl/rl/agent/temporal_correlated_agent.py:25,343-366,
 * black_box_agent.py:25,124-127; generate_minibatches / select_batch,
 * mprl/util/util_data_structure.py:362-391).  row_index int64 [R]: the epoch's
 * permutation of the rows, drawn by the caller with numpy's global generator
 * exactly as the reference draws it (np.random.shuffle of arange(R)); it is cut
 * like np.array_split into num_minibatches consecutive pieces (the first
 * R % num_minibatches pieces one row longer).  Per piece, in order, ONE optimizer
 * step: forward + value loss (mean over the piece) + backward over the GATHERED
 * rows (x, returns, old_values are read through the index in place -- nothing is
 * copied), slab reduction, Adam.  stats: float [num_minibatches][4], zeroed by
 * the caller: per piece {mean loss, |grad|^2, |grad|, |grad| clipped} ([2], [3]
 * are written only with grad_clip > 0 or an exchange; otherwise the caller takes
 * the root of [1]).  adam_step: the step count INCLUDING the first piece's
 * update; grad_clip: clip_grad_norm (<= 0: none; needs num_params <= 2^17);
 * the other arguments as in tce_mlp_critic_f32.  tce_mlpw_critic_minibatch_*:
 * the same for the wide / float64 value nets (declared below). */
int tce_mlp_critic_minibatch_f32(
    const float* x, int64_t env_stride, int64_t row_stride, int T, int64_t R, int din,
    const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
    const float* b3, int act, const float* returns, const float* old_values, float clip,
    const int64_t* row_index, int num_minibatches, float* partials, float* grad, float* stats,
    int max_workgroups, float* adam_param, float* adam_m, float* adam_v, float* adam_state,
    float lr, float beta1, float beta2, float eps, float weight_decay, float adam_step,
    float grad_clip, float grad_scale, void* xchg, void* stream);

/* ---- fused critic epoch for wide / double-precision value networks ---------
 * The contract of tce_mlp_critic_f32 for  D_in -> hidden -> hidden -> 1  with
 * hidden = 256 (float32; float64 with D_in <= 24) or hidden = 128 (float64):
 * the critics of mprl/config/box_push_random_init/tcp/entire/shared.yaml:7,95-96
 * (float64, 256 x 2) and mprl/config/table_tennis_4d/tcp/entire/shared.yaml:98-103,
 * on the exact matrix instructions (v_mfma_f32_16x16x4_f32 /
 * v_mfma_f64_16x16x4_f64).  tce_mlpw_supported(D_in, hidden, element size) tells
 * whether a combination is built.  workspace: tce_mlpw_workspace_len(R, hidden,
 * backward) elements (W2 images; backward: H1, dY2, dY1 of all rows, written by
 * the chain kernel and contracted by the weight-gradient kernel); partials:
 * [tce_mlpw_grid(), tce_mlpw_num_params(D_in, hidden) + 2]; grad in the order
 * W1, b1, W2, b2, w3, b3; stats[2] = {mean loss, |grad|^2} zeroed by the caller.
 * partials == NULL: forward only (values required).
 */
int tce_mlpw_supported(int din, int hidden, int elem_size);
int tce_mlpw_grid(void);
int64_t tce_mlpw_num_params(int din, int hidden);
int64_t tce_mlpw_workspace_len(int64_t R, int hidden, int backward);
int tce_mlpw_critic_f32(const float* x, int64_t env_stride, int64_t row_stride, int T,
                        int64_t R, int din, int hidden, const float* w1, const float* b1,
                        const float* w2, const float* b2, const float* w3, const float* b3,
                        int act, const float* returns, const float* old_values, float clip,
                        float* values, float* workspace, float* partials, float* grad,
                        float* stats, int max_workgroups, float* adam_param, float* adam_m,
                        float* adam_v, float* adam_state, float lr, float beta1, float beta2,
                        float eps, float weight_decay, float adam_step, float grad_scale,
                        void* xchg, void* stream);
int tce_mlpw_critic_f64(const double* x, int64_t env_stride, int64_t row_stride, int T,
                        int64_t R, int din, int hidden, const double* w1, const double* b1,
                        const double* w2, const double* b2, const double* w3,
                        const double* b3, int act, const double* returns,
                        const double* old_values, double clip, double* values,
                        double* workspace, double* partials, double* grad, double* stats,
                        int max_workgroups, double* adam_param, double* adam_m,
                        double* adam_v, double* adam_state, double lr, double beta1,
                        double beta2, double eps, double weight_decay, double adam_step,
                        double grad_scale, void* xchg, void* stream);
int tce_mlpw_critic_minibatch_f32(
    const float* x, int64_t env_stride, int64_t row_stride, int T, int64_t R, int din,
    int hidden, const float* w1, const float* b1, const float* w2, const float* b2,
    const float* w3, const float* b3, int act, const float* returns, const float* old_values,
    float clip, const int64_t* row_index, int num_minibatches, float* workspace,
    float* partials, float* grad, float* stats, int max_workgroups, float* adam_param,
    float* adam_m, float* adam_v, float* adam_state, float lr, float beta1, float beta2,
    float eps, float weight_decay, float adam_step, float grad_clip, float grad_scale,
    void* xchg, void* stream);
int tce_mlpw_critic_minibatch_f64(
    const double* x, int64_t env_stride, int64_t row_stride, int T, int64_t R, int din,
    int hidden, const double* w1, const double* b1, const double* w2, const double* b2,
    const double* w3, const double* b3, int act, const double* returns,
    const double* old_values, double clip, const int64_t* row_index, int num_minibatches,
    double* workspace, double* partials, double* grad, double* stats, int max_workgroups,
    double* adam_param, double* adam_m, double* adam_v, double* adam_state, double lr,
    double beta1, double beta2, double eps, double weight_decay, double adam_step,
    double grad_clip, double grad_scale, void* xchg, void* stream);

/* The backward launch of tce_mlp_critic_f32 (same buffers and contract;
 * partials != NULL required) on the f16 matrix cores with SPLIT operands: every
 * fp32 operand is carried as hi = f16(x), lo = f16((x - hi) 2^11) and every
 * product as hi*hi + 2^-11 (hi*lo + lo*hi) accumulated in fp32 -- 22+ significand
 * bits per operand, results within fp32 summation-order noise of the exact-fp32
 * kernel (tests/test_mlp16_gpu.py) at 3/16 of its matrix-core cycles.  Operands
 * must lie inside the f16 range (|x| < 65504); larger inputs give inf / nan in
 * stats[0], which the caller checks as it does for the fp32 kernel. */
int tce_mlp_critic_f16x2(const float* x, int64_t env_stride, int64_t row_stride, int T,
                         int64_t R, int din, const float* w1, const float* b1,
                         const float* w2, const float* b2, const float* w3, const float* b3,
                         int act, const float* returns, const float* old_values, float clip,
                         float* values, float* partials, float* grad, float* stats,
                         int max_workgroups, float* adam_param, float* adam_m, float* adam_v,
                         float* adam_state, float lr, float beta1, float beta2, float eps,
                         float weight_decay, float adam_step, float grad_scale, void* xchg,
                         void* stream);

/* The backward launch of tce_mlp_critic_f32 (same buffers and contract;
 * partials != NULL required) on the bf16 matrix cores with THREE-PART operands:
 * every fp32 operand x is carried as b0 = bf16(x), b1 = bf16(x - b0), b2 =
 * bf16(x - b0 - b1) -- x = b0 + b1 + b2 exactly: 24 significand bits, the
 * exponent range of fp32, no range restriction -- and every product as its six
 * partial products of order <= 2 (a0 b0, a0 b1, a1 b0, a1 b1, a0 b2, a2 b0),
 * each exact in the matrix core, accumulated in fp32.  The dropped terms are
 * below 2^-25 of the product (35 x under the rounding noise of an fp32 FMA
 * chain): loss and gradients are as close to the fp64 truth as the exact-fp32
 * kernel's (tests/test_mlpb_gpu.py) at 6/16 of its matrix-core cycles
 * (csrc/mlpb.hip).  Replaces the same reference lines as tce_mlp_critic_f32
 * (mprl/rl/agent/temporal_correlated_agent.py:343-366, 688-716). */
int tce_mlp_critic_bf16x3(const float* x, int64_t env_stride, int64_t row_stride, int T,
                          int64_t R, int din, const float* w1, const float* b1,
                          const float* w2, const float* b2, const float* w3, const float* b3,
                          int act, const float* returns, const float* old_values, float clip,
                          float* values, float* partials, float* grad, float* stats,
                          int max_workgroups, float* adam_param, float* adam_m, float* adam_v,
                          float* adam_state, float lr, float beta1, float beta2, float eps,
                          float weight_decay, float adam_step, float grad_scale, void* xchg,
                          void* stream);

/* ---- policy mean net on rows (float32 / float64, one or two hidden layers) ----
 * D_in <= 64 -> hidden (-> hidden) -> D_out <= 64 over N rows, torch Linear
 * layout, parameters FLAT in the order of MLP.parameters()
 * (mprl/util/util_nn.py:75-160): W1 [hidden][D_in] | b1 | (W2 [hidden][hidden]
 * | b2) | W3 [D_out][hidden] | b3.  The mean nets of
 * mprl/config/box_push_random_init/tcp/entire/shared.yaml:7,75-78 (float64,
 * 128 x 2) and mprl/config/table_tennis_4d/tcp/entire/shared.yaml:78-81
 * (256 x 1, tanh) -- MLP.forward (util_nn.py:225-246) and what autograd does
 * for it inside update_policy (temporal_correlated_agent.py:523-612) -- on the
 * exact 16x16x4 matrix instructions.  act: 0 tanh, 1 relu, 2 leaky_relu, 3
 * softplus (util_nn.py:16-25); no output activation.
 * tce_pmlp_supported: hidden 128 with 1 or 2 hidden layers, hidden 256 with 1,
 * element size 4 or 8.
 * forward: out [N][D_out]; h1 / h2 (nullable; h2 only with two hidden layers)
 * [N][hidden] receive the hidden activations the backward needs.
 * backward: grad [P] = the gradient of sum(grad_out * out) w.r.t. every
 * parameter (P = tce_pmlp_num_params); partials: [tce_pmlp_max_slabs()][P]
 * scratch (per-workgroup slabs, added in a fixed order).  param, h1, h2 must be
 * 16-byte aligned; x rows may be strided. */
int tce_pmlp_supported(int din, int hidden, int num_hidden, int dout, int elem_size);
int64_t tce_pmlp_num_params(int din, int hidden, int num_hidden, int dout);
int tce_pmlp_max_slabs(void);
int tce_pmlp_forward_f32(const float* x, int64_t x_stride, int64_t N, int din, int hidden,
                         int num_hidden, int dout, int act, const float* param, float* h1,
                         float* h2, float* out, void* stream);
int tce_pmlp_forward_f64(const double* x, int64_t x_stride, int64_t N, int din, int hidden,
                         int num_hidden, int dout, int act, const double* param, double* h1,
                         double* h2, double* out, void* stream);
int tce_pmlp_backward_f32(const float* x, int64_t x_stride, int64_t N, int din, int hidden,
                          int num_hidden, int dout, int act, const float* param, const float* h1,
                          const float* h2, const float* grad_out, float* partials, float* grad,
                          void* stream);
int tce_pmlp_backward_f64(const double* x, int64_t x_stride, int64_t N, int din, int hidden,
                          int num_hidden, int dout, int act, const double* param,
                          const double* h1, const double* h2, const double* grad_out,
                          double* partials, double* grad, void* stream);

/* One full-batch epoch of a value function D_in -> H (-> H) -> 1 of the shapes
 * above (the black-box agent's 256 x 1 critic of
 * mprl/config/table_tennis_4d/bbrl/entire/shared.yaml:90-91):
 * mprl/rl/agent/black_box_agent.py:128-146 -- critic(states), value_loss
 * (:438-466, clip_critic > 0: old_values required), backward, grad_norm_clip,
 * Adam -- as forward, one loss kernel, backward, one clip + Adam launch.
 * param / grad / m / v: FLAT in MLP.parameters() order (P <= 2^17); step: the
 * optimizer's step count INCLUDING this update; ws: [tce_pmlp_critic_ws_len(N,
 * hidden)] elements, ZEROED once by the caller (the loss kernel re-arms its
 * ticket); partials: [tce_pmlp_max_slabs()][P]; rec_row3 = {loss, |g|, |g|
 * clipped}.  do_adam == 0: the caller steps.  xchg (nullable; needs do_adam): env
 * shards -- the Adam launch adds the peers' gradients first (tce_xchg_adam_*),
 * grad_scale = 1 / world. */
int64_t tce_pmlp_critic_ws_len(int64_t N, int hidden);
int tce_pmlp_critic_epoch_f32(const float* x, int64_t x_stride, const float* returns,
                              const float* old_values, int64_t N, int din, int hidden,
                              int num_hidden, int act, float clip_critic, float* param,
                              float* grad, float* m, float* v, float* opt_state, float lr,
                              float beta1, float beta2, float eps, float weight_decay,
                              float clip_grad, float grad_scale, int do_adam, float step,
                              float* ws, float* partials, float* rec_row3, void* xchg,
                              void* stream);
int tce_pmlp_critic_epoch_f64(const double* x, int64_t x_stride, const double* returns,
                              const double* old_values, int64_t N, int din, int hidden,
                              int num_hidden, int act, double clip_critic, double* param,
                              double* grad, double* m, double* v, double* opt_state, double lr,
                              double beta1, double beta2, double eps, double weight_decay,
                              double clip_grad, double grad_scale, int do_adam, double step,
                              double* ws, double* partials, double* rec_row3, void* xchg,
                              void* stream);

/* ---- small two-hidden-layer networks (black-box agent) ----------------------
 * D_in <= 64 -> H -> H -> D_out (H in {32, 64}, D_out <= 64), fp32, torch
 * Linear layout, parameters FLAT in the order W1 [H][D_in] | b1 | W2 [H][H] |
 * b2 | W3 [D_out][H] | b3 (the order of MLP.parameters(),
 * mprl/util/util_nn.py:75-160): the networks of
 * mprl/config/metaworld/bbrl/entire/shared.yaml:61-91.  act: 0 tanh, 1 relu,
 * 2 leaky_relu, 3 softplus (mprl/util/util_nn.py:16-25).  head: 0 forward,
 * 1 value loss, 2 black-box policy objective (what must fit the LDS).
 * ws: ZEROED ONCE by the caller, float [tce_smlp_ws_len(N, din, H, dout)]
 * (gradient slabs + ticket, re-armed by every launch).
 *
 * tce_smlp_forward: out [N][dout] = MLP(x) (MLP.forward, util_nn.py:225-246).
 *
 * tce_smlp_critic_epochs: `epochs` full-batch critic epochs of
 * BlackBoxAgent.update_critic (mprl/rl/agent/black_box_agent.py:105-157), TWO
 * launches each: the row kernel (forward, value loss -- clip_critic > 0: the
 * clipped form of :391-419 --, backward, per-workgroup gradient slabs) and the
 * slab reduction into `grad` [P], which -- do_adam and clip_grad <= 0 -- also
 * applies the Adam step of torch.optim.Adam(lr, weight_decay)
 * (mprl/rl/agent/abstract_agent.py:62-82) on param / m / v [P] (opt_state[0] =
 * step count; first_step = count INCLUDING the first of these epochs); with
 * clip_grad > 0 (mprl/util/util_numerical.py:244-275) tce_adam_flat follows
 * instead.  do_adam == 0 (the caller steps itself) requires epochs == 1.
 * xchg (nullable, both entries; needs do_adam): the envs are sharded over the
 * ranks of this exchange (tce_xchg_*) -- the launch that applies Adam (critic:
 * tce_xchg_adam behind the slab reduction; policy: the one-workgroup finish
 * kernel) adds the peers' gradients in rank order first and takes grad_scale =
 * 1 / world, so the `epochs` epochs of a sharded update are the same ONE call;
 * grad keeps the SUMMED gradient, the critic's rec rows are then {mean loss of
 * the local shard, |g|, |g| clipped} of the rank-averaged gradient.
 * rec [epochs][3], ZEROED by the caller, receives {mean loss, |grad|^2, -}.
 *
 * tce_bb_policy_epochs: `epochs` policy epochs of BlackBoxAgent.update_policy
 * (black_box_agent.py:159-389) for a shared (non-contextual) covariance:
 * Cholesky head (tce_chol_build_fwd on the variance vector param + P, nvec
 * entries), covariance projection, K x K KL parts, ONE row kernel (mean net
 * forward, mean projection, log-prob of `actions` under the projected
 * Gaussian, surrogate, trust region loss, their gradients, mean net backward),
 * slab reduction, tce_kl_cov_proj_bwd, finish (Cholesky head backward into
 * grad [P, P + nvec), clip, Adam on all P + nvec entries, record) -- seven
 * launches; diag & 1 (std_only factors, beta == NULL: every shipped BBRL
 * config): the K x K steps collapse into K-vector steps, four launches; diag & 2:
 * `mats` still holds L_old^-1 from an earlier call with the same L_old (the
 * per-epoch calls of a balance-check iteration skip its recomputation).  rec
 * [epochs][rec_stride], rec_stride 7 = {surrogate, entropy loss, trust region
 * loss, total, entropy, |g|, |g| clipped}, >= 19: followed by the 12 means of
 * kl_old_new_proj (black_box_agent.py:391-436) = {mean, cov, shape, volume}
 * difference of (new || old), (new || proj), (proj || old).  mats: float [tce_bb_policy_mats_len(K)], holds after
 * the call L_new | L_proj | (scratch) in [K,K] blocks of pitch (K*K rounded up
 * to 4); proj_ctx: double [tce_kl_cov_proj_ctx_len(K)], zeroed by the caller
 * once per update; mean_new_out / proj_mean_out (nullable) [N,K]: the last
 * epoch's means. */
/* Generic dense layer on the exact 16x16x4 matrix instructions (csrc/glin.hip)
 * for MLP shapes none of the fused families covers -- the reference sizes its
 * nets from arbitrary YAML numbers (mprl/util/util_hyperparams.py:8-46) and the
 * contextual covariance head is a second MLP with K (K + 1) / 2 outputs
 * (mprl/rl/policy/abstract_policy.py:96-109).  1 <= D_in, D_out <=
 * tce_glin_max_dim() (4096).  forward: y [R][D_out] = x W^T + bias (x rows at
 * x_stride elements, W torch Linear layout [D_out][D_in], bias nullable; the
 * activation stays with the caller).  backward: grad_x [R][D_in] = grad_y W
 * (nullable), grad_W [D_out][D_in] = grad_y^T x (the rows split over the chip,
 * partial tiles summed in a fixed order), grad_b [D_out] = column sums of grad_y
 * (nullable); ws: tce_glin_ws_len(R, D_in, D_out) elements. */
int tce_glin_max_dim(void);
int64_t tce_glin_ws_len(int64_t R, int din, int dout);
int tce_glin_forward_f32(const float* x, int64_t x_stride, int64_t R, int din, int dout,
                         const float* W, const float* bias, float* y, void* stream);
int tce_glin_forward_f64(const double* x, int64_t x_stride, int64_t R, int din, int dout,
                         const double* W, const double* bias, double* y, void* stream);
int tce_glin_backward_f32(const float* x, int64_t x_stride, const float* grad_y, const float* W,
                          int64_t R, int din, int dout, float* grad_x, float* grad_W,
                          float* grad_b, float* ws, void* stream);
int tce_glin_backward_f64(const double* x, int64_t x_stride, const double* grad_y,
                          const double* W, int64_t R, int din, int dout, double* grad_x,
                          double* grad_W, double* grad_b, double* ws, void* stream);
/* Skinny linear layer on rows, y [N][dout] = x [N][din] A (+ bias): transposed
 * != 0: A = W^T for a torch Linear weight W [dout][din] (MLP.forward's output
 * layer, mprl/util/util_nn.py:225-246); transposed == 0: A = W for W
 * [din][dout] (its input gradient dX = dY W under autograd).  D_in <= 256,
 * D_out <= 128; x rows may be strided, y is contiguous; bias nullable. */
int tce_lin_rows_f32(const float* x, int64_t x_stride, int64_t N, int din, int dout,
                     const float* W, int transposed, const float* bias, float* y, void* stream);
int tce_smlp_supported(int din, int H, int dout, int head);
int64_t tce_smlp_num_params(int din, int H, int dout);
int64_t tce_smlp_ws_len(int64_t N, int din, int H, int dout);
int64_t tce_bb_policy_mats_len(int K);
int tce_smlp_forward_f32(const float* x, int64_t x_stride, int64_t N, int din, int H, int dout,
                         int act, const float* param, float* out, void* stream);
int tce_smlp_critic_epochs_f32(const float* x, int64_t x_stride, const float* returns,
                               const float* old_values, int64_t N, int din, int H, int act,
                               float clip_critic, float* param, float* grad, float* m, float* v,
                               float* opt_state, float lr, float beta1, float beta2, float eps,
                               float weight_decay, float clip_grad, float grad_scale, int do_adam,
                               int first_step, int epochs, float* ws, float* rec, void* xchg,
                               void* stream);
int tce_bb_policy_epochs_f32(const float* x, int64_t x_stride, const float* actions,
                             const float* logp_old, const float* adv, const float* mean_old,
                             const float* L_old, int64_t N, int din, int H, int K, int act,
                             int nvec, float min_std, float eps_mean, double eps_cov,
                             const float* beta, int entropy_eq, float tr_coeff,
                             int tr_include_cov, float ent_coef, float* param, float* grad,
                             float* m, float* v, float* opt_state, float lr, float beta1,
                             float beta2, float eps, float weight_decay, float clip_grad,
                             float grad_scale, int do_adam, int diag, int epochs,
                             double* proj_ctx, float* ws, float* mats, float* rec,
                             int rec_stride, float* mean_new_out, float* proj_mean_out,
                             void* xchg, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TCE_HIP_H */
