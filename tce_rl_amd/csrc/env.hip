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
// env are a serial recurrence whose per-step latency IS the kernel time (all
// 4096 waves are resident at once): the hand position and |qd|^2 reach the
// other lanes by v_readlane / DPP (an LDS round trip per step cost 2 300 cycles
// of dependent latency, 474 us per episode whatever was stored), the desired
// trajectory is staged in LDS a block of 16 steps ahead and read one step
// ahead; 16 waves per CU keep the stores in flight.
//
// Observation row: [q(dof) | qd(dof) | obj(3) | goal(3) | 0 ... | time |
//                   des_pos(dof) | des_vel(dof)],   D = d_task + 1 + 2 dof.
#include "common.h"

#ifndef ENV_SKIP
#define ENV_SKIP 0      // diagnostic builds (scripts/time_env.py): 1 no state stores, 2 no moments, 4 no reward stores
#endif

namespace {

enum { FAM_REACH = 0, FAM_PUSH = 1, FAM_TABLE_TENNIS = 2, FAM_HOPPER = 3 };

// value of lane `src` (wave-uniform index) in every lane: v_readlane, no LDS
__device__ inline float lane_bcast(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
__device__ inline double lane_bcast(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
// sum over lanes 0..7 of the wave (valid in lanes 0..7)
__device__ inline float sum8_lo(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
  return v;
}
__device__ inline double sum8_lo(double v) { return dpp_sum8(v); }

template <typename real>
__global__ __launch_bounds__(64) void env_rollout_kernel(
    const real* __restrict__ actions, const real* __restrict__ init_obs, int family,
    int T, int dof, int d_task, real dt, real kp, real kd,
    real* __restrict__ states, real* __restrict__ rewards,
    uint8_t* __restrict__ flags, real* __restrict__ metrics,
    const real* __restrict__ shift, double* __restrict__ partials) {
  const int64_t n = blockIdx.x;
  const int c = threadIdx.x;
  const int D = d_task + 1 + 2 * dof;
  const real* o0 = init_obs + n * D;
  const real* act = actions + n * (int64_t)T * 2 * dof;
  // lane roles: c < dof integrates degree of freedom c and owns the columns c
  // (q) and dof + c (qd) of the row; lanes 2 dof .. D - 1 own one column each
  // (object, goal, padding, time, desired pos / vel); the rest idles
  const bool dyn = c < dof;
  const bool own = c >= 2 * dof && c < D;
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
  // moments of my column(s) (shifted by the running mean), row 0 = initial obs
  const int cb = dyn ? dof + c : c;                 // second column of a dyn lane
  const double k = (shift && (dyn || own)) ? (double)shift[c] : 0.0;
  const double kb = (shift && dyn) ? (double)shift[cb] : 0.0;
  double m1 = 0, m2 = 0, m1b = 0, m2b = 0;
  real* srow = states ? states + n * (int64_t)(T + 1) * D : nullptr;
  if (dyn || own) {
    const real x0 = o0[c];
    if (srow) srow[c] = x0;
    const double d0 = (double)x0 - k;
    m1 = d0;
    m2 = d0 * d0;
  }
  if (dyn) {
    const real x0 = o0[cb];
    if (srow) srow[cb] = x0;
    const double d0 = (double)x0 - kb;
    m1b = d0;
    m2b = d0 * d0;
  }
  // desired trajectory: blocks of BS steps (BS * 2 dof <= 256 floats) come in
  // with ONE coalesced load per wave, issued a whole block ahead, and are
  // parked in a wave-private LDS slab; a step's values are read from it one
  // step ahead of their use.  (Per-step global loads put an L2 round trip,
  // ~750 ns under load, on every step's critical path.)
  constexpr int BS = 16;
  const int A = 2 * dof;
  __shared__ real slab[2][BS * 16];
  typedef real ld4 __attribute__((ext_vector_type(4), aligned(sizeof(real))));
  const int64_t total = (int64_t)T * A;
  auto fetch_block = [&](int blk) -> ld4 {
    ld4 v = {0, 0, 0, 0};
    const int64_t e0 = (int64_t)blk * BS * A + 4 * c;
    if (4 * c < BS * A) {
      if (e0 + 3 < total) v = *reinterpret_cast<const ld4*>(act + e0);
      else
        for (int j = 0; j < 4; ++j)
          if (e0 + j < total) v[j] = act[e0 + j];
    }
    return v;
  };
  auto park_block = [&](int buf, ld4 v) {
    if (4 * c < BS * A) *reinterpret_cast<ld4*>(&slab[buf][4 * c]) = v;
  };
  const int nblk = (T + BS - 1) / BS;
  park_block(0, fetch_block(0));
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  real dist2 = 0;
  // values of step 0
  real dp_n = dyn ? slab[0][c] : real(0), dv_n = dyn ? slab[0][dof + c] : real(0);
  real ac_n = has_a ? slab[0][acol] : real(0);
  for (int blk = 0; blk < nblk; ++blk) {
   const int buf = blk & 1;
   const int steps = T - blk * BS < BS ? T - blk * BS : BS;
   // the next block: loaded and parked here, in one piece (the wait for it
   // also drains this wave's outstanding stores -- loads and stores share one
   // in-order counter -- but only once per BS steps; a load result kept in
   // registers across the steps makes the compiler wait at every step)
   if (blk + 1 < nblk) park_block(buf ^ 1, fetch_block(blk + 1));
   asm volatile("" ::: "memory");
   __builtin_amdgcn_wave_barrier();
   for (int u = 0; u < steps; ++u) {
    const int i = blk * BS + u;
    const real dp_i = dp_n, dv_i = dv_n, ac_i = ac_n;
    {                                                // read step i + 1 from the slab
      const int un = u + 1 < steps ? u + 1 : 0;
      const real* sb = slab[u + 1 < steps ? buf : buf ^ 1] + un * A;
      if (dyn) { dp_n = sb[c]; dv_n = sb[dof + c]; }
      if (has_a) ac_n = sb[acol];
    }
    if (dyn) {                                       // PD-tracked point mass
      const real a = kp * (dp_i - q) + kd * (dv_i - qd);
      qd = qd + dt * a;
      q = q + dt * qd;
    }
    // hand = q[:3] and |qd|^2 to every lane: lane broadcasts / DPP, no LDS
    real h[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) h[j] = lane_bcast(q, j);
    const real v2 = lane_bcast(sum8_lo(dyn ? qd * qd : real(0)), 0);
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
        for (int j = 0; j < 3; ++j) ov[j] = lane_bcast(qd, j);
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
    real* orow = srow ? srow + (int64_t)(i + 1) * D : nullptr;
    if (dyn || own) {
      real x;
      if (dyn) x = q;
      else if (c < 2 * dof + 3) x = c == 2 * dof ? obj[0] : (c == 2 * dof + 1 ? obj[1] : obj[2]);
      else if (c < 2 * dof + 6) x = c == 2 * dof + 3 ? goal[0] : (c == 2 * dof + 4 ? goal[1] : goal[2]);
      else if (c < d_task) x = 0;
      else if (c == d_task) x = t;
      else x = ac_i;
#if !(ENV_SKIP & 1)
      if (orow) orow[c] = x;
#endif
#if !(ENV_SKIP & 2)
      const double d = (double)x - k;
      m1 += d;
      m2 += d * d;
#else
      m1 += (double)x;
#endif
    }
    if (dyn) {
#if !(ENV_SKIP & 1)
      if (orow) orow[cb] = qd;
#endif
#if !(ENV_SKIP & 2)
      const double d = (double)qd - kb;
      m1b += d;
      m2b += d * d;
#endif
    }
#if !(ENV_SKIP & 4)
    if (c == 0) {
      rewards[n * (int64_t)T + i] = rew;
      if (flags) flags[n * (int64_t)T + i] = event ? 1 : 0;
    }
#else
    if (c == 0 && i == T - 1) rewards[n * (int64_t)T + i] = rew;
#endif
   }
  }
  if (c == 0 && metrics) {
    const real lim = family == FAM_TABLE_TENNIS ? real(0.09) : real(0.0025);
    const bool ok = dist2 < lim && (family != FAM_TABLE_TENNIS || event);
    metrics[2 * n] = ok ? real(1) : real(0);
    metrics[2 * n + 1] = sqrt(dist2);
  }
  if (partials) {
    if (dyn || own) {
      partials[(n * D + c) * 2 + 0] = m1;
      partials[(n * D + c) * 2 + 1] = m2;
    }
    if (dyn) {
      partials[(n * D + cb) * 2 + 0] = m1b;
      partials[(n * D + cb) * 2 + 1] = m2b;
    }
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
    TCE_CHECK_ARG(dof >= 3 && dof <= 8, "env_rollout: 3 <= dof <= 8");           \
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
