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
// Mapping: 16 lanes (one DPP row) per env, 4 envs per wave; lanes < dof
// integrate one degree of freedom each; the other columns that change with the
// step (object, time, desired row) are packed one per lane, the goal / padding
// columns are written into the LDS staging rows once (closed-form moments);
// rows leave in 16-step blocks as a fixed number of 16-byte stores.  The T steps of an env are a serial
// recurrence whose per-step latency IS the kernel time (every wave is resident):
// the hand position and |qd|^2 reach the row's lanes by ds_bpermute / DPP (an
// LDS memory round trip per step cost 2 300 cycles of dependent latency), the
// desired trajectory is staged in LDS a block of 16 steps ahead and read one
// step ahead.  (One wave per env -- round-2's first version -- was
// instruction-bound: 200 instructions per step and env, 17 % of the HBM rate;
// this one reaches 24 %, 139 of its 243 us being the bare recurrence.)
//
// Observation row: [q(dof) | qd(dof) | obj(3) | goal(3) | 0 ... | time |
//                   des_pos(dof) | des_vel(dof)],   D = d_task + 1 + 2 dof.
#include "common.h"

#ifndef ENV_SKIP
#define ENV_SKIP 0      // diagnostic builds (scripts/time_env.py): 1 no state stores, 2 no moments, 4 no reward stores
#endif

namespace {

enum { FAM_REACH = 0, FAM_PUSH = 1, FAM_TABLE_TENNIS = 2, FAM_HOPPER = 3 };

// sum over the 16 lanes of a DPP row (all 16 get it)
__device__ inline float row_sum16(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
  return v;
}
__device__ inline double row_sum16(double v) {
  v = dpp_sum8(v);
  v += dpp_perm_f64<0x140>(v);
  return v;
}
// value of lane `src` of my 16-lane row
__device__ inline float row_get(float v, int src_lane) { return __shfl(v, src_lane, 64); }
__device__ inline double row_get(double v, int src_lane) { return __shfl(v, src_lane, 64); }

constexpr int ENV_LPE = 16;            // lanes per env (one DPP row)
constexpr int ENV_EPW = 64 / ENV_LPE;  // envs per wave

// NQ: 16-byte pieces per lane and block of the staged rows (>= BS * D / 64);
// NQ = 0: D is not a multiple of 4, per-step 4-byte column stores instead
template <typename real, int NQ>
__global__ __launch_bounds__(64) void env_rollout_kernel(
    const real* __restrict__ actions, const real* __restrict__ init_obs, int family, int64_t N,
    int T, int dof, int d_task, real dt, real kp, real kd,
    real* __restrict__ states, real* __restrict__ rewards,
    uint8_t* __restrict__ flags, real* __restrict__ metrics,
    const real* __restrict__ shift, double* __restrict__ partials) {
  const int c = threadIdx.x;
  const int g = c & (ENV_LPE - 1), e = c / ENV_LPE;  // lane in the env's row, env of the wave
  const int64_t n_raw = (int64_t)blockIdx.x * ENV_EPW + e;
  const bool live = n_raw < N;
  const int64_t n = live ? n_raw : N - 1;            // dead rows shadow the last env, never store
  const int rowbase = c & ~(ENV_LPE - 1);
  const int D = d_task + 1 + 2 * dof;
  const int A = 2 * dof;
  const real* o0 = init_obs + n * D;
  const real* act = actions + n * (int64_t)T * A;
  // lane roles inside an env's row: g < dof integrates degree of freedom g and
  // owns the columns g (q) and dof + g (qd); every lane also owns the columns
  // 2 dof + g + 16 m < D (object, goal, padding, time, desired pos / vel)
  const bool dyn = g < dof;
  real q = dyn ? o0[g] : real(0), qd = dyn ? o0[dof + g] : real(0);
  real obj[3], goal[3], ov[3], hp[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    obj[j] = o0[2 * dof + j];
    goal[j] = o0[2 * dof + 3 + j];
    hp[j] = o0[j];
    ov[j] = family == FAM_TABLE_TENNIS ? -obj[j] / (real(T) * dt) : real(0);
  }
  bool event = false;
  // Columns that change with the step besides q / qd: object (3), time, desired
  // [pos | vel] (2 dof) -- lane g owns number g + 16 md of that list (kind 0..2
  // obj, 7 time, 8 + k element k of the desired row; -1 none).  The goal and
  // padding columns never change: they are written into the staging rows once
  // and their moments are a closed form at the end.
  constexpr int ENV_MD = 2, ENV_MC = 4;
  int dcol[ENV_MD], dkind[ENV_MD];
  double dk_[ENV_MD];
#pragma unroll
  for (int md = 0; md < ENV_MD; ++md) {
    const int j = g + ENV_LPE * md;
    int col = -1, kind = -1;
    if (j < 3) { col = 2 * dof + j; kind = j; }
    else if (j == 3) { col = d_task; kind = 7; }
    else if (j - 4 < A) { col = d_task + 1 + (j - 4); kind = 8 + (j - 4); }
    dcol[md] = col;
    dkind[md] = kind;
    dk_[md] = (shift && col >= 0) ? (double)shift[col] : 0.0;
  }
  int ccol[ENV_MC];
#pragma unroll
  for (int m = 0; m < ENV_MC; ++m) {
    const int col = 2 * dof + 3 + g + ENV_LPE * m;
    ccol[m] = col < d_task ? col : -1;
  }
  const double kq = (shift && dyn) ? (double)shift[g] : 0.0;
  const double kv = (shift && dyn) ? (double)shift[dof + g] : 0.0;
  double mq1 = 0, mq2 = 0, mv1 = 0, mv2 = 0, md1[ENV_MD], md2[ENV_MD];
  real* srow = states ? states + n * (int64_t)(T + 1) * D : nullptr;
  // row 0 = the reset observation
  if (dyn) {
    const real x0 = o0[g], x1 = o0[dof + g];
    if (srow && live) { srow[g] = x0; srow[dof + g] = x1; }
    mq1 = (double)x0 - kq; mq2 = mq1 * mq1;
    mv1 = (double)x1 - kv; mv2 = mv1 * mv1;
  }
#pragma unroll
  for (int md = 0; md < ENV_MD; ++md) {
    md1[md] = 0; md2[md] = 0;
    if (dcol[md] >= 0) {
      const real x0 = o0[dcol[md]];
      if (srow && live) srow[dcol[md]] = x0;
      md1[md] = (double)x0 - dk_[md]; md2[md] = md1[md] * md1[md];
    }
  }
#pragma unroll
  for (int m = 0; m < ENV_MC; ++m)
    if (ccol[m] >= 0 && srow && live) srow[ccol[m]] = o0[ccol[m]];
  // desired trajectory: blocks of BS steps, one coalesced load per env block,
  // parked in an LDS slab (per wave: 4 envs x BS x 2 dof), read one step ahead
  constexpr int BS = 16;
  __shared__ real slab[2][ENV_EPW][BS * 16];
  // the rows of a block of BS steps are staged here and leave as consecutive
  // 16-byte pieces once per block (lane = piece: BS * D * s contiguous bytes per
  // env) -- per-step 4-byte column stores were bound by the number of store
  // instructions (8 per step and wave), not by bytes
  __shared__ real stage[ENV_EPW][BS * 64];
  __shared__ real rstage[ENV_EPW][BS];
  __shared__ uint8_t fstage[ENV_EPW][BS];
  constexpr bool wide = NQ > 0;                      // D % 4 == 0: rows start on 16-byte boundaries
  if (wide) {
#pragma unroll
    for (int m = 0; m < ENV_MC; ++m)
      if (ccol[m] >= 0) {
        const real xc = o0[ccol[m]];
        for (int u = 0; u < BS; ++u) stage[e][u * D + ccol[m]] = xc;
      }
  }
  typedef real ld4 __attribute__((ext_vector_type(4), aligned(sizeof(real))));
  const int64_t total = (int64_t)T * A;
  // the env's block = BS * A <= 256 contiguous elements: 16 lanes x 4 x 4
  struct Blk { ld4 v[4]; };
  // (pieces past BS * A are never parked, so they stay unwritten; elements past
  // the episode's end repeat its last one -- clamped, unconditional loads: a
  // register written on some paths only makes the compiler wait for every
  // outstanding memory operation at the top of the block loop)
  auto fetch_block = [&](int blk, Blk& b) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int el = 4 * (g + ENV_LPE * j);
      int64_t e0 = (int64_t)blk * BS * A + (el < BS * A ? el : 0);
      // a chunk that would run past the episode loads its last 4 elements and
      // is SHIFTED so that element e0 + k stays in slot k (T * A = 2 mod 4 --
      // odd T with an odd dof -- ends in a half chunk; a plain clamp put
      // act[total - 4 ..] into the slots of act[total - 2 ..])
      const int64_t ec = e0 + 3 < total ? e0 : total - 4;
      const int sh = (int)(e0 - ec) < 3 ? (int)(e0 - ec) : 3;
      const ld4 v = *reinterpret_cast<const ld4*>(act + ec);
      ld4 w;
      w[0] = sh == 0 ? v[0] : sh == 1 ? v[1] : sh == 2 ? v[2] : v[3];
      w[1] = sh == 0 ? v[1] : sh == 1 ? v[2] : v[3];
      w[2] = sh == 0 ? v[2] : v[3];
      w[3] = v[3];
      b.v[j] = w;
    }
  };
  auto park_block = [&](int buf, const Blk& b) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int el = 4 * (g + ENV_LPE * j);
      if (el < BS * A) *reinterpret_cast<ld4*>(&slab[buf][e][el]) = b.v[j];
    }
  };
  const int nblk = (T + BS - 1) / BS;
  Blk nextb;
  fetch_block(0, nextb);
  park_block(0, nextb);
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  real dist2 = 0;
  // values of step 0: dp / dv for the dof lanes, the action elements of owned columns
  real dp_n = dyn ? slab[0][e][g] : real(0), dv_n = dyn ? slab[0][e][dof + g] : real(0);
  real oa_n[ENV_MD];
#pragma unroll
  for (int md = 0; md < ENV_MD; ++md) oa_n[md] = dkind[md] >= 8 ? slab[0][e][dkind[md] - 8] : real(0);
  for (int blk = 0; blk < nblk; ++blk) {
   const int buf = blk & 1;
   const int steps = T - blk * BS < BS ? T - blk * BS : BS;
   // the next block's loads are issued here and waited for after the BS steps,
   // BEFORE this block's rows are flushed: loads and stores share one in-order
   // counter, a load issued behind the flush would wait for the stores to
   // reach memory (several us under load, once per block); the steps in
   // between touch no global memory
   if (blk + 1 < nblk) fetch_block(blk + 1, nextb);
   for (int u = 0; u < steps; ++u) {
    const int i = blk * BS + u;
    const real dp_i = dp_n, dv_i = dv_n;
    real oa_i[ENV_MD];
#pragma unroll
    for (int md = 0; md < ENV_MD; ++md) oa_i[md] = oa_n[md];
    if (u + 1 < steps) {                             // read step i + 1 from the slab
      const real* sb = slab[buf][e] + (u + 1) * A;
      if (dyn) { dp_n = sb[g]; dv_n = sb[dof + g]; }
#pragma unroll
      for (int md = 0; md < ENV_MD; ++md)
        if (dkind[md] >= 8) oa_n[md] = sb[dkind[md] - 8];
    }
    if (dyn) {                                       // PD-tracked point mass
      const real a = kp * (dp_i - q) + kd * (dv_i - qd);
      qd = qd + dt * a;
      q = q + dt * qd;
    }
    // hand = q[:3] and |qd|^2 to every lane of the env's row
    real h[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) h[j] = row_get(q, rowbase + j);
    const real v2 = row_sum16(dyn ? qd * qd : real(0));
    const real t = real(i + 1) * dt;
    real rew;
    if (family == FAM_PUSH) {
      real c2 = 0;
#pragma unroll
      for (int j = 0; j < 3; ++j) c2 += (hp[j] - obj[j]) * (hp[j] - obj[j]);
      const bool touch = c2 < real(0.01);            // in contact: carried along
#pragma unroll
      for (int j = 0; j < 3; ++j) obj[j] += touch ? h[j] - hp[j] : real(0);
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
      const bool hit = !event && b2 < real(0.04);    // racket meets the ball
      real qv[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) qv[j] = row_get(qd, rowbase + j);
#pragma unroll
      for (int j = 0; j < 3; ++j) ov[j] = hit ? qv[j] : ov[j];
      event = event || hit;
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
    real* orow = (srow && live) ? srow + (int64_t)(i + 1) * D : nullptr;
    if (dyn) {
#if !(ENV_SKIP & 1)
      if (wide) { stage[e][u * D + g] = q; stage[e][u * D + dof + g] = qd; }
      else if (orow) { orow[g] = q; orow[dof + g] = qd; }
#endif
#if !(ENV_SKIP & 2)
      const double d1 = (double)q - kq, d2 = (double)qd - kv;
      mq1 += d1; mq2 += d1 * d1;
      mv1 += d2; mv2 += d2 * d2;
#endif
    }
#pragma unroll
    for (int md = 0; md < ENV_MD; ++md) {
      if (dkind[md] >= 0) {
        const int kd_ = dkind[md];
        real x;
        if (kd_ < 3) x = kd_ == 0 ? obj[0] : (kd_ == 1 ? obj[1] : obj[2]);
        else if (kd_ == 7) x = t;
        else x = oa_i[md];
#if !(ENV_SKIP & 1)
        if (wide) stage[e][u * D + dcol[md]] = x;
        else if (orow) orow[dcol[md]] = x;
#endif
#if !(ENV_SKIP & 2)
        const double d = (double)x - dk_[md];
        md1[md] += d; md2[md] += d * d;
#endif
      }
    }
    if (!wide && orow) {                             // the never-changing columns
#pragma unroll
      for (int m = 0; m < ENV_MC; ++m)
        if (ccol[m] >= 0) orow[ccol[m]] = o0[ccol[m]];
    }
#if !(ENV_SKIP & 4)
    if (g == 0) {
      rstage[e][u] = rew;
      fstage[e][u] = event ? 1 : 0;
    }
#else
    if (g == 0 && live && i == T - 1) rewards[n * (int64_t)T + i] = rew;
#endif
   }
   // ---- park the next block and take its first step's values
   if (blk + 1 < nblk) {
     park_block(buf ^ 1, nextb);
     asm volatile("" ::: "memory");
     __builtin_amdgcn_wave_barrier();
     const real* sb = slab[buf ^ 1][e];
     if (dyn) { dp_n = sb[g]; dv_n = sb[dof + g]; }
#pragma unroll
     for (int md = 0; md < ENV_MD; ++md)
       if (dkind[md] >= 8) oa_n[md] = sb[dkind[md] - 8];
   }
   // ---- flush the block: rows i0 + 1 .. i0 + steps of the state buffer,
   // rewards and flags of steps i0 .. i0 + steps - 1
   asm volatile("" ::: "memory");
   __builtin_amdgcn_wave_barrier();
   {
     // a FIXED number of unconditional stores per block (pieces past the end
     // repeat the last one; rows of dead envs repeat env N - 1's, same values):
     // the compiler can then count them (s_waitcnt vmcnt(n)) and the next
     // block's loads do not wait for these stores to reach memory
     const int i0 = blk * BS;
#if !(ENV_SKIP & 1)
     if (wide && srow) {
       real* dst = srow + (int64_t)(i0 + 1) * D;
       const int n4 = steps * D / 4;
       typedef real st4 __attribute__((ext_vector_type(4), aligned(4 * sizeof(real))));
#pragma unroll
       for (int qq = 0; qq < (NQ > 0 ? NQ : 1); ++qq) {
         int p4 = g + ENV_LPE * qq;
         p4 = p4 < n4 ? p4 : n4 - 1;
         *reinterpret_cast<st4*>(dst + 4 * p4) = *reinterpret_cast<const st4*>(&stage[e][4 * p4]);
       }
     }
#endif
#if !(ENV_SKIP & 4)
     {
       const int gg = g < steps ? g : steps - 1;
       rewards[n * (int64_t)T + i0 + gg] = rstage[e][gg];
       if (flags) flags[n * (int64_t)T + i0 + gg] = fstage[e][gg];
     }
#endif
   }
   asm volatile("" ::: "memory");
   __builtin_amdgcn_wave_barrier();
  }
  if (g == 0 && live && metrics) {
    const real lim = family == FAM_TABLE_TENNIS ? real(0.09) : real(0.0025);
    const bool ok = dist2 < lim && (family != FAM_TABLE_TENNIS || event);
    metrics[2 * n] = ok ? real(1) : real(0);
    metrics[2 * n + 1] = sqrt(dist2);
  }
  if (partials && live) {
    if (dyn) {
      partials[(n * D + g) * 2 + 0] = mq1;
      partials[(n * D + g) * 2 + 1] = mq2;
      partials[(n * D + dof + g) * 2 + 0] = mv1;
      partials[(n * D + dof + g) * 2 + 1] = mv2;
    }
#pragma unroll
    for (int md = 0; md < ENV_MD; ++md)
      if (dcol[md] >= 0) {
        partials[(n * D + dcol[md]) * 2 + 0] = md1[md];
        partials[(n * D + dcol[md]) * 2 + 1] = md2[md];
      }
#pragma unroll
    for (int m = 0; m < ENV_MC; ++m)
      if (ccol[m] >= 0) {                              // T + 1 equal rows
        const double d = (double)o0[ccol[m]] - (shift ? (double)shift[ccol[m]] : 0.0);
        partials[(n * D + ccol[m]) * 2 + 0] = (double)(T + 1) * d;
        partials[(n * D + ccol[m]) * 2 + 1] = (double)(T + 1) * d * d;
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
    TCE_CHECK_ARG((int64_t)T * 2 * dof >= 4, "env_rollout: T * 2 dof >= 4");       \
    const int need = (16 * (d_task + 1 + 2 * dof) + 63) / 64;                      \
    const dim3 grid((unsigned)((N + ENV_EPW - 1) / ENV_EPW));                      \
    hipStream_t st = (hipStream_t)stream;                                          \
    if ((d_task + 1 + 2 * dof) % 4 != 0)                                           \
      hipLaunchKernelGGL((env_rollout_kernel<REAL, 0>), grid, dim3(64), 0, st,     \
                         actions, init_obs, family, N, T, dof, d_task, dt, kp, kd, \
                         states, rewards, event_flags, metrics, shift,             \
                         moment_partials);                                         \
    else if (need <= 6)                                                            \
      hipLaunchKernelGGL((env_rollout_kernel<REAL, 6>), grid, dim3(64), 0, st,     \
                         actions, init_obs, family, N, T, dof, d_task, dt, kp, kd, \
                         states, rewards, event_flags, metrics, shift,             \
                         moment_partials);                                         \
    else if (need <= 9)                                                            \
      hipLaunchKernelGGL((env_rollout_kernel<REAL, 9>), grid, dim3(64), 0, st,     \
                         actions, init_obs, family, N, T, dof, d_task, dt, kp, kd, \
                         states, rewards, event_flags, metrics, shift,             \
                         moment_partials);                                         \
    else if (need <= 12)                                                           \
      hipLaunchKernelGGL((env_rollout_kernel<REAL, 12>), grid, dim3(64), 0, st,    \
                         actions, init_obs, family, N, T, dof, d_task, dt, kp, kd, \
                         states, rewards, event_flags, metrics, shift,             \
                         moment_partials);                                         \
    else                                                                           \
      hipLaunchKernelGGL((env_rollout_kernel<REAL, 16>), grid, dim3(64), 0, st,    \
                         actions, init_obs, family, N, T, dof, d_task, dt, kp, kd, \
                         states, rewards, event_flags, metrics, shift,             \
                         moment_partials);                                         \
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
