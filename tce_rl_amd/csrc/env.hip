// GPU-resident synthetic env suite: one launch = one whole episode
// of N batched envs speaking the fancy_gym-TCE step protocol
// (mprl/rl/sampler/temporal_correlated_sampler.py:226-303): PD-tracked
// point-mass dynamics following the desired [pos | vel] trajectory, per-family
// task logic (reach / push / table-tennis-like / hopper-jump-like), and in the
// SAME pass the whole `step_states` buffer [N, T+1, D] (initial observation in
// row 0), the step rewards, the event flags of make_mdp_reward, the task
// metrics and the per-column moment partials of the observation running
// mean/std (RunningMeanStd.update, mprl/util/util_numerical.py:315-337) --
// the buffer is written once and never re-read for its statistics.
//
// HBM-bound: writes (T+1)*D*s B and reads T*2*dof*s B per env (C2: 394 + 66 MB).
// Mapping: one wave per env, lane = observation column (coalesced D*s-byte row
// stores); lanes < dof integrate one degree of freedom each; the T steps of an
// env are a serial recurrence, the loads of step i+1 are issued before step i
// is computed; 4096 envs = 16 waves per CU keep the stores in flight.
//
// Observation row: [q(dof) | qd(dof) | obj(3) | goal(3) | 0 ... | time |
//                   des_pos(dof) | des_vel(dof)],   D = d_task + 1 + 2 dof.
#include "common.h"

namespace {

enum { FAM_REACH = 0, FAM_PUSH = 1, FAM_TABLE_TENNIS = 2, FAM_HOPPER = 3 };

template <typename real>
__global__ __launch_bounds__(64) void env_rollout_kernel(
    const real* __restrict__ actions, const real* __restrict__ init_obs, int family,
    int T, int dof, int d_task, real dt, real kp, real kd,
    real* __restrict__ states, real* __restrict__ rewards,
    uint8_t* __restrict__ flags, real* __restrict__ metrics,
    const real* __restrict__ shift, double* __restrict__ partials) {
  __shared__ real row[64];
  const int64_t n = blockIdx.x;
  const int c = threadIdx.x;
  const int D = d_task + 1 + 2 * dof;
  const real* o0 = init_obs + n * D;
  const real* act = actions + n * (int64_t)T * 2 * dof;
  const bool dyn = c < dof;
  const int acol = c - d_task - 1;                  // my column inside the action
  const bool has_a = acol >= 0 && c < D;
  real q = dyn ? o0[c] : real(0), qd = dyn ? o0[dof + c] : real(0);
  real obj[3], goal[3], ov[3], hp[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    obj[j] = o0[2 * dof + j];
    goal[j] = o0[2 * dof + 3 + j];
    hp[j] = o0[j];
    ov[j] = family == FAM_TABLE_TENNIS ? -obj[j] / (real(T) * dt) : real(0);
  }
  bool event = false;
  // moments of my column (shifted by the running mean), row 0 = initial obs
  const double k = (shift && c < D) ? (double)shift[c] : 0.0;
  double m1 = 0, m2 = 0;
  if (c < D) {
    const real x0 = o0[c];
    if (states) states[n * (int64_t)(T + 1) * D + c] = x0;
    const double d0 = (double)x0 - k;
    m1 = d0;
    m2 = d0 * d0;
  }
  // loads of step 0
  real dp = dyn ? act[c] : real(0), dv = dyn ? act[dof + c] : real(0);
  real ac = has_a ? act[acol] : real(0);
  real dist2 = 0;
  for (int i = 0; i < T; ++i) {
    const real dp_i = dp, dv_i = dv, ac_i = ac;
    if (i + 1 < T) {                                 // prefetch step i + 1
      const real* nx = act + (int64_t)(i + 1) * 2 * dof;
      if (dyn) { dp = nx[c]; dv = nx[dof + c]; }
      if (has_a) ac = nx[acol];
    }
    if (dyn) {                                       // PD-tracked point mass
      const real a = kp * (dp_i - q) + kd * (dv_i - qd);
      qd = qd + dt * a;
      q = q + dt * qd;
      row[c] = q;
      row[dof + c] = qd;
    }
    __syncthreads();
    real h[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) h[j] = row[j];
    real v2 = 0;
    for (int d = 0; d < dof; ++d) v2 += row[dof + d] * row[dof + d];
    const real t = real(i + 1) * dt;
    real rew;
    if (family == FAM_PUSH) {
      real c2 = 0;
#pragma unroll
      for (int j = 0; j < 3; ++j) c2 += (hp[j] - obj[j]) * (hp[j] - obj[j]);
      if (c2 < real(0.01)) {                         // in contact: carried along
#pragma unroll
        for (int j = 0; j < 3; ++j) obj[j] += h[j] - hp[j];
      }
      real g2 = 0, o2 = 0;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        g2 += (obj[j] - goal[j]) * (obj[j] - goal[j]);
        o2 += (h[j] - obj[j]) * (h[j] - obj[j]);
      }
      dist2 = g2;
      rew = -g2 - real(0.1) * o2 - real(1e-3) * v2;
    } else if (family == FAM_TABLE_TENNIS) {
#pragma unroll
      for (int j = 0; j < 3; ++j) obj[j] += dt * ov[j];
      real b2 = 0;
#pragma unroll
      for (int j = 0; j < 3; ++j) b2 += (h[j] - obj[j]) * (h[j] - obj[j]);
      if (!event && b2 < real(0.04)) {               // racket meets the ball
        event = true;
#pragma unroll
        for (int j = 0; j < 3; ++j) ov[j] = row[dof + j];
      }
      real g2 = 0;
#pragma unroll
      for (int j = 0; j < 2; ++j) g2 += (obj[j] - goal[j]) * (obj[j] - goal[j]);
      dist2 = g2;
      rew = (event ? -g2 : -b2) - real(1e-3) * v2;
    } else {                                         // reach / hopper-jump-like
      real g2 = 0;
#pragma unroll
      for (int j = 0; j < 3; ++j) g2 += (h[j] - goal[j]) * (h[j] - goal[j]);
      dist2 = g2;
      rew = -g2 - real(1e-3) * v2;
      if (family == FAM_HOPPER && h[2] > real(0.3)) event = true;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) hp[j] = h[j];
    if (c < D) {
      real x;
      if (c < 2 * dof) x = row[c];
      else if (c < 2 * dof + 3) x = c == 2 * dof ? obj[0] : (c == 2 * dof + 1 ? obj[1] : obj[2]);
      else if (c < 2 * dof + 6) x = c == 2 * dof + 3 ? goal[0] : (c == 2 * dof + 4 ? goal[1] : goal[2]);
      else if (c < d_task) x = 0;
      else if (c == d_task) x = t;
      else x = ac_i;
      if (states) states[(n * (int64_t)(T + 1) + i + 1) * D + c] = x;
      const double d = (double)x - k;
      m1 += d;
      m2 += d * d;
    }
    if (c == 0) {
      rewards[n * (int64_t)T + i] = rew;
      if (flags) flags[n * (int64_t)T + i] = event ? 1 : 0;
    }
    __syncthreads();                                 // row is rewritten next step
  }
  if (c == 0 && metrics) {
    const real lim = family == FAM_TABLE_TENNIS ? real(0.09) : real(0.0025);
    const bool ok = dist2 < lim && (family != FAM_TABLE_TENNIS || event);
    metrics[2 * n] = ok ? real(1) : real(0);
    metrics[2 * n + 1] = sqrt(dist2);
  }
  if (partials && c < D) {
    partials[(n * D + c) * 2 + 0] = m1;
    partials[(n * D + c) * 2 + 1] = m2;
  }
}

// merge the moment partials of one batch into the running statistics
// (update_from_moments, util_numerical.py:321-337); same arithmetic as
// rms_finalize_kernel of rollout.hip, one workgroup per column
template <typename real>
__global__ __launch_bounds__(256) void env_rms_finalize_kernel(
    const double* __restrict__ partials, int64_t nparts, int D,
    const real* shift /* may alias mean */, double batch_count, double count,
    real* mean, real* var) {
  __shared__ double red[4];
  const int c = blockIdx.x;
  double t1 = 0, t2 = 0;
  for (int64_t i = threadIdx.x; i < nparts; i += 256) {
    t1 += partials[(i * D + c) * 2 + 0];
    t2 += partials[(i * D + c) * 2 + 1];
  }
  t1 = block_sum(t1, red);
  t2 = block_sum(t2, red);
  if (threadIdx.x != 0) return;
  const double k = shift ? (double)shift[c] : 0.0;
  const double n = batch_count;
  const double b_mean = k + t1 / n;
  const double b_var = n > 1 ? (t2 - t1 * t1 / n) / (n - 1.0) : (double)NAN;
  const double m = (double)mean[c], v = (double)var[c];
  const double delta = b_mean - m;
  const double tot = count + n;
  mean[c] = (real)(m + delta * n / tot);
  var[c] = (real)((v * count + b_var * n + delta * delta * count * n / tot) / tot);
}

}  // namespace

extern "C" {

#define DEFINE_ENV(SFX, REAL)                                                      \
  int tce_env_rollout_##SFX(const REAL* actions, const REAL* init_obs, int family, \
                            int64_t N, int T, int dof, int d_task, REAL dt,        \
                            REAL kp, REAL kd, REAL* states, REAL* rewards,         \
                            uint8_t* event_flags, REAL* metrics,                   \
                            const REAL* shift, double* moment_partials,            \
                            void* stream) {                                        \
    TCE_CHECK_ARG(actions && init_obs && rewards && N > 0 && T > 0,                \
                  "env_rollout: null buffer / empty batch");                       \
    TCE_CHECK_ARG(family >= 0 && family <= 3, "env_rollout: unknown env family");  \
    TCE_CHECK_ARG(dof >= 3 && dof <= 16, "env_rollout: 3 <= dof <= 16");           \
    TCE_CHECK_ARG(d_task >= 2 * dof + 6 && d_task + 1 + 2 * dof <= 64,             \
                  "env_rollout: 2 dof + 6 <= d_task and D <= 64");                 \
    TCE_CHECK_ARG(N < (1ll << 31), "env_rollout: too many envs");                  \
    hipLaunchKernelGGL(env_rollout_kernel<REAL>, dim3((unsigned)N), dim3(64), 0,   \
                       (hipStream_t)stream, actions, init_obs, family, T, dof,     \
                       d_task, dt, kp, kd, states, rewards, event_flags, metrics,  \
                       shift, moment_partials);                                    \
    TCE_LAUNCH_CHECK();                                                            \
    return 0;                                                                      \
  }                                                                                \
  int tce_rms_merge_##SFX(const double* moment_partials, int64_t nparts, int D,    \
                          const REAL* shift, double batch_count, double count,     \
                          REAL* mean, REAL* var, void* stream) {                   \
    TCE_CHECK_ARG(moment_partials && mean && var && nparts > 0 && D > 0 &&         \
                      batch_count > 0,                                             \
                  "rms_merge: bad arguments");                                     \
    hipLaunchKernelGGL(env_rms_finalize_kernel<REAL>, dim3(D), dim3(256), 0,       \
                       (hipStream_t)stream, moment_partials, nparts, D, shift,     \
                       batch_count, count, mean, var);                             \
    TCE_LAUNCH_CHECK();                                                            \
    return 0;                                                                      \
  }

DEFINE_ENV(f32, float)
DEFINE_ENV(f64, double)

}  // extern "C"
