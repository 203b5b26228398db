// GAE reverse scan + segment advantage for gfx950 (MI355X).
//
// Replaces the Python reverse loop of
//   mprl/rl/agent/temporal_correlated_agent.py:118-181 (get_advantage_return)
// and the one-hot einsum of
//   mprl/rl/agent/temporal_correlated_agent.py:236-286 (value_subtraction).
//
// Layout: rewards/dones/adv/ret [N, T] row-major (env x step), values
// [N, T+1].  HBM-bound: 18 B per (env, step) in fp32.
//
// gae_dpp_kernel -- "systolic" scan on DPP row shifts, no LDS, no barriers.
// The recurrence x_t = a_t + k_t * x_{t+1} is serial in t and we keep the
// reference's exact operation order (bit-identical fp32/fp64 results), so a
// lone wave pays one dependent mul + add per step whatever we do; what can be
// removed is everything else.  A wave64 is 4 DPP rows of 16 lanes: row r owns
// env 4*wave + r, lane c of the row owns the 4 steps 64*p + 4*c .. +3 of pass
// p (one 16-byte load / store per array per lane: 256 contiguous bytes per
// row, the VMEM instruction count is what bounds a CU here).  Every lane
// evaluates its 4 steps from the x of lane c+1 (neighbour read folded into
// the multiply as a row_shl:1 DPP operand) 16 times; lane c is final after
// iteration 15 - c because its input became final one iteration earlier, so
// after 16 iterations the row holds the exact sequential result.  The carry
// into lane 15 (x of the previous pass' lane 0) is folded into its last a
// with the same mul + add.  All operands sit in the owning lane's registers,
// loaded straight from global memory; the loads of a whole 512-step tile are
// issued up front (one wave per SIMD may use the whole register file) and
// each pass waits only for its own, so HBM streaming overlaps the chain.
#include "common.h"
#include <stdlib.h>

// Keep mul and add un-fused: the scan must round exactly like the reference's
// separate torch ops.
#pragma clang fp contract(off)

namespace {

#ifdef GAE_STAMP
__device__ unsigned long long* g_stamps;
#endif

constexpr int GAE_ROW = 16;          // lanes per DPP row
constexpr int GAE_VEC = 4;           // steps per lane per pass
constexpr int GAE_PSTEPS = GAE_ROW * GAE_VEC;   // 64 steps per pass
constexpr int GAE_MAXP = 8;          // passes per tile -> 512 steps
constexpr int GAE_ENVS_PER_WAVE = 4;
constexpr int GAE_BT = 256;          // 4 independent waves per workgroup
constexpr int GAE_ENVS_PER_BLOCK = GAE_ENVS_PER_WAVE * GAE_BT / 64;

constexpr int DPP_ROW_SHL1 = 0x101;  // lane c <- lane c+1 (lane 15: bound)
constexpr int DPP_ROW_ROR15 = 0x12F; // lane c <- lane (c+1) % 16

template <int CTRL, bool BOUND_ZERO>
__device__ inline float dpp_mov(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(
      __float_as_int(old), __float_as_int(v), CTRL, 0xf, 0xf, BOUND_ZERO));
}
template <int CTRL, bool BOUND_ZERO>
__device__ inline double dpp_mov(double old, double v) {
  const long long o = __double_as_longlong(old), s = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp((int)o, (int)s, CTRL, 0xf, 0xf, BOUND_ZERO);
  const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(s >> 32), CTRL,
                                             0xf, 0xf, BOUND_ZERO);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// 4 consecutive elements of a row.  Rows are only element-aligned (pitches T
// and T+1); the hardware handles the unaligned 16-byte access.  Both forms are
// branch-free (hipcc waits vmcnt(0) wherever two load paths merge):
//   VEC  (T % 4 == 0): one 16-byte access at a base clamped into the row; a
//        lane is then either wholly inside the row or wholly past it;
//   VEC + RAG (T % 4 != 0, T >= 4; round 6: table tennis's T = 350): the same
//        16-byte access; the ONE lane of a row that straddles its end gets the
//        clamped base's elements shifted into place by selects (elements past
//        the end are never used: the caller masks them), and stores its valid
//        elements one by one;
//   !VEC (T < 4): four element accesses with clamped indices.
template <typename E>
struct Vec4 { E v[4]; };
template <bool VEC, bool RAG, typename E>
__device__ inline Vec4<E> load4(const E* __restrict__ row, int t, int last) {
  typedef E vec __attribute__((ext_vector_type(4), aligned(sizeof(E))));
  Vec4<E> o;
  if (VEC) {
    const int base = min(t, last - 3);
    const vec x = *reinterpret_cast<const vec*>(row + base);
    if (!RAG) {
      o.v[0] = x.x; o.v[1] = x.y; o.v[2] = x.z; o.v[3] = x.w;
    } else {
      const int s = t - base;                  // 0 inside the row, 1 .. 3 for the straddling lane
      o.v[0] = s == 0 ? x.x : (s == 1 ? x.y : (s == 2 ? x.z : x.w));
      o.v[1] = s == 0 ? x.y : (s == 1 ? x.z : x.w);
      o.v[2] = s == 0 ? x.z : x.w;
      o.v[3] = x.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) o.v[i] = row[min(t + i, last)];
  }
  return o;
}
template <bool VEC, bool RAG, typename E>
__device__ inline void store4(E* __restrict__ row, int t, int T, const E* x) {
  typedef E vec __attribute__((ext_vector_type(4), aligned(sizeof(E))));
  if (VEC) {
    if (RAG ? t + 3 < T : t < T) {
      vec o = {x[0], x[1], x[2], x[3]};
      *reinterpret_cast<vec*>(row + t) = o;
    } else if (RAG) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (t + i < T) row[t + i] = x[i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (t + i < T) row[t + i] = x[i];
  }
}

// NP_STATIC > 0: the row fits one tile of exactly NP_STATIC passes (T <= 512):
// every pass guard folds away at compile time, so the code has no branches
// around loads and hipcc emits counted vmcnt waits (a wave-uniform branch
// around a load makes it wait vmcnt(0) at the join, which serialises the
// stream).  NP_STATIC == 0: generic runtime pass count / multi-tile path.
template <typename real, bool USE_GAE, bool VEC, int NP_STATIC, int PF = 4, bool RAG = false>
__global__ __launch_bounds__(GAE_BT) void gae_dpp_kernel(
    const real* __restrict__ rewards, const real* __restrict__ values,
    const uint8_t* __restrict__ dones, const uint8_t* __restrict__ tl_dones,
    real* __restrict__ adv, real* __restrict__ ret, int64_t N, int T,
    real gamma, real lam) {
#ifdef GAE_STAMP
#define STAMP(i) if ((threadIdx.x & 63) == 0) g_stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_amdgcn_s_memtime()
#else
#define STAMP(i)
#endif
  STAMP(0);
  const int lane = threadIdx.x & 63;
  const int row = lane >> 4, col = lane & 15;
  const int64_t wave = (int64_t)blockIdx.x * (GAE_BT / 64) + (threadIdx.x >> 6);
  const int64_t env_raw = wave * GAE_ENVS_PER_WAVE + row;
  if (wave * GAE_ENVS_PER_WAVE >= N) return;          // whole wave idle
  const bool env_ok = env_raw < N;
  const int64_t env = env_ok ? env_raw : N - 1;       // clamped, always valid
  const real* rrow = rewards + env * (int64_t)T;
  const real* vrow = values + env * (int64_t)(T + 1);
  const uint8_t* drow = dones + env * (int64_t)T;
  const uint8_t* lrow = tl_dones + env * (int64_t)T;
  real* arow = adv + env * (int64_t)T;
  real* orow = ret + env * (int64_t)T;

  constexpr int TILE = GAE_MAXP * GAE_PSTEPS;
  real xfin = USE_GAE ? real(0) : vrow[T];   // lane 0 holds x entering lane 15
  const int ntiles = NP_STATIC > 0 ? 1 : (T + TILE - 1) / TILE;
  for (int tile = ntiles - 1; tile >= 0; --tile) {
    const int t0 = tile * TILE;
    const int np = NP_STATIC > 0 ? NP_STATIC
                                 : min(GAE_MAXP, (T - t0 + GAE_PSTEPS - 1) / GAE_PSTEPS);
    // ---- software pipeline: the loads of pass p - PF are issued before the
    // chain of pass p, so HBM streaming runs PF passes ahead of the chain
    // (PF = 4: with ONE wave per SIMD -- 4096 envs = 1024 waves -- issuing the
    // whole tile up front only fills the memory queues and stalls the in-order
    // wave before it can start computing; PF = 8 = the whole tile: with several
    // waves per SIMD -- 32768 envs -- the other waves compute meanwhile and the
    // deeper queue is what a cold HBM stream needs, gae_launch picks)
    Vec4<real> rv[GAE_MAXP], vv[GAE_MAXP];
    Vec4<uint8_t> dn[GAE_MAXP], tl[GAE_MAXP];
    // V just above the tile (V_{t+1} of its last step) and V_T
    const real vtop = vrow[min(t0 + np * GAE_PSTEPS, T)];
    const real vT = vrow[T];
    STAMP(1);
#pragma unroll
    for (int pp = GAE_MAXP - 1 + PF; pp >= 0; --pp) {
      {
        const int q = pp - PF;                         // pass to prefetch
        if (q >= 0 && q < GAE_MAXP && q < np) {        // wave-uniform
          const int qi = q < 0 ? 0 : (q >= GAE_MAXP ? GAE_MAXP - 1 : q);
          const int t = t0 + q * GAE_PSTEPS + col * GAE_VEC;
          vv[qi] = load4<VEC, RAG>(vrow, t, T - 1);   // V_T comes from vT
          rv[qi] = load4<VEC, RAG>(rrow, t, T - 1);
          dn[qi] = load4<VEC, RAG>(drow, t, T - 1);
          tl[qi] = load4<VEC, RAG>(lrow, t, T - 1);
        }
      }
      const int p = pp < GAE_MAXP ? pp : GAE_MAXP - 1;
      if (pp < GAE_MAXP && p < np) {                   // wave-uniform
        const int t = t0 + p * GAE_PSTEPS + col * GAE_VEC;
        // V_{t+1} of this lane's last step: lane c+1's first V; lane 15 takes
        // the first V of the pass above (or V just above the tile); the lane
        // that ends the row takes V_T
        const real vabove = (p + 1 < np) ? vv[(p + 1 < GAE_MAXP) ? p + 1 : p].v[0] : vtop;
        real vn3 = dpp_mov<DPP_ROW_SHL1, false>(
            dpp_mov<DPP_ROW_ROR15, true>(real(0), vabove), vv[p].v[0]);
        if (t + 4 >= T) vn3 = vT;
        real a[4], k[4], c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          real vnext = (i < 3) ? vv[p].v[(i < 3) ? i + 1 : 3] : vn3;
          if ((!VEC || RAG) && t + i + 1 >= T) vnext = vT;
          const real nd = dn[p].v[i] ? real(0) : real(1);
          const real ntl = tl[p].v[i] ? real(0) : real(1);
          const real disc = gamma * nd;
          c[i] = 0;
          if (USE_GAE) {
            const real dv = disc * vnext;
            const real tmp = rv[p].v[i] + dv;
            const real td = tmp - vv[p].v[i];
            const real kk = disc * lam;
            a[i] = td * ntl;          // mask in {0,1}: exact products
            k[i] = kk * ntl;
          } else {
            a[i] = rv[p].v[i] * ntl;
            k[i] = disc * ntl;
            c[i] = (real(1) - ntl) * vv[p].v[i];
          }
          if (t + i >= T) { a[i] = 0; k[i] = 1; c[i] = 0; }   // identity past T
        }
        // carry into lane 15's last step: a' = a + k * x_{t+1}
        const real cin = dpp_mov<DPP_ROW_ROR15, true>(real(0), xfin);
        const real kc = k[3] * cin;
        const real ap = a[3] + kc;
        if (col == GAE_ROW - 1) a[3] = ap;
        real x[4] = {0, 0, 0, 0};
#pragma unroll
        for (int it = 0; it < GAE_ROW; ++it) {
          const real xin = dpp_mov<DPP_ROW_SHL1, true>(real(0), x[0]);
          real kx, s;
          kx = k[3] * xin;  s = a[3] + kx; if (!USE_GAE) s = s + c[3]; x[3] = s;
          kx = k[2] * x[3]; s = a[2] + kx; if (!USE_GAE) s = s + c[2]; x[2] = s;
          kx = k[1] * x[2]; s = a[1] + kx; if (!USE_GAE) s = s + c[1]; x[1] = s;
          kx = k[0] * x[1]; s = a[0] + kx; if (!USE_GAE) s = s + c[0]; x[0] = s;
        }
        xfin = x[0];
        if (env_ok) {
          real rt[4], av[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            rt[i] = USE_GAE ? (x[i] + vv[p].v[i]) : x[i];
            av[i] = rt[i] - vv[p].v[i];
          }
          store4<VEC, RAG>(orow, t, T, rt);
          store4<VEC, RAG>(arow, t, T, av);
        }
      }
    }
    STAMP(4);
  }
}

// Un-normalised `value_subtraction` segment advantage + per-block moment
// partials.  One wave per env, lane = pair:
//   A = sum_{t in [a,b)} gamma^t r_t / gamma^a + gamma^(b-a) V_b - V_a
// rewards / values were just streamed by the scan and are L2 / MALL resident.
template <typename real>
__global__ __launch_bounds__(256) void segadv_vs_kernel(
    const real* __restrict__ rewards, const real* __restrict__ values,
    const int64_t* __restrict__ pairs, int P, real* __restrict__ seg_out,
    double* __restrict__ partials, int64_t N, int T, real gamma) {
  __shared__ double s_red[4];
  const int lane = threadIdx.x & 63;
  const int64_t env0 = (int64_t)blockIdx.x * 4;
  double loc = 0, loc2 = 0;
  int cnt = 0;
  // pass 1: values + local sum; pass 2 (registers): M2 around the block mean
  real outv[4];          // up to 4 pair-chunks of 64 per lane (P <= 256)
  const int64_t env = env0 + (threadIdx.x >> 6);
  const bool env_ok = env < N;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int p = q * 64 + lane;
    outv[q] = 0;
    if (env_ok && p < P) {
      const int a = (int)pairs[2 * p], b = (int)pairs[2 * p + 1];
      const real* rrow = rewards + env * (int64_t)T;
      const real* vrow = values + env * (int64_t)(T + 1);
      // batches of 8 independent loads (clamped), summed in step order
      real acc = 0, out;
      const real va = vrow[a], vb = vrow[b];
      for (int t = a; t < b; t += 8) {
        real x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = rrow[min(t + i, b - 1)];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (t + i < b)
            acc += (gamma == real(1)) ? x[i] : x[i] * pow(gamma, real(t + i));
        }
      }
      if (gamma == real(1)) out = acc + vb - va;
      else out = acc / pow(gamma, real(a)) + pow(gamma, real(b - a)) * vb - va;
      seg_out[env * (int64_t)P + p] = out;
      outv[q] = out;
      loc += (double)out;
      ++cnt;
    }
  }
  const double ncnt = block_sum((double)cnt, s_red);
  const double mean = block_sum(loc, s_red) / ncnt;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int p = q * 64 + lane;
    if (env_ok && p < P) {
      const double d = (double)outv[q] - mean;
      loc2 += d * d;
    }
  }
  const double m2 = block_sum(loc2, s_red);
  if (threadIdx.x == 0) {
    partials[3 * (int64_t)blockIdx.x + 0] = ncnt;
    partials[3 * (int64_t)blockIdx.x + 1] = mean;
    partials[3 * (int64_t)blockIdx.x + 2] = m2;
  }
}

// (count, mean, M2) partials of an arbitrary array, one triple per block.
template <typename real>
__global__ __launch_bounds__(256) void moments_partial_kernel(
    const real* __restrict__ x, int64_t n, double* __restrict__ partials) {
  __shared__ double s_red[4];
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = blockIdx.x * per;
  const int64_t hi = tmin<int64_t>(n, lo + per);
  const double cnt = (double)tmax<int64_t>(hi - lo, 0);
  double loc = 0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) loc += (double)x[i];
  const double tot = block_sum(loc, s_red);
  const double mean = cnt > 0 ? tot / cnt : 0.0;
  double loc2 = 0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const double d = (double)x[i] - mean;
    loc2 += d * d;
  }
  const double m2 = block_sum(loc2, s_red);
  if (threadIdx.x == 0) {
    partials[3 * (int64_t)blockIdx.x + 0] = cnt;
    partials[3 * (int64_t)blockIdx.x + 1] = mean;
    partials[3 * (int64_t)blockIdx.x + 2] = m2;
  }
}

// Combine partial (count, mean, M2) triples -> stats = {count, mean, M2}.
// mean = sum n_b mean_b / n ; M2 = sum [M2_b + n_b (mean_b - mean)^2] (exact).
__global__ __launch_bounds__(256) void moments_finalize_kernel(
    const double* __restrict__ partials, int nparts, double* __restrict__ stats) {
  __shared__ double s_red[4];
  double n = 0, s = 0;
  for (int i = threadIdx.x; i < nparts; i += 256) {
    n += partials[3 * i];
    s += partials[3 * i] * partials[3 * i + 1];
  }
  const double ntot = block_sum(n, s_red);
  const double mean = ntot > 0 ? block_sum(s, s_red) / ntot : 0.0;
  double m2 = 0;
  for (int i = threadIdx.x; i < nparts; i += 256) {
    const double d = partials[3 * i + 1] - mean;
    m2 += partials[3 * i + 2] + partials[3 * i] * d * d;
  }
  const double m2tot = block_sum(m2, s_red);
  if (threadIdx.x == 0) {
    stats[0] = ntot;
    stats[1] = mean;
    stats[2] = m2tot;
  }
}

// y = clamp((x - mean) / (std_unbiased + eps), -clip, clip); clip <= 0: none.
// stats == nullptr: clamp only.
template <typename real>
__global__ __launch_bounds__(256) void normalize_kernel(
    const real* __restrict__ x, real* __restrict__ y, int64_t n,
    const double* __restrict__ stats, real eps, real clip, int single_std_one) {
  real mean = 0, denom = 1;
  if (stats) {
    const double cnt = stats[0];
    mean = (real)stats[1];
    // reference: torch .std() (unbiased); BBRL guards len==1 with std := 1
    real sd;
    if (cnt <= 1.0) sd = single_std_one ? real(1) : real(NAN);
    else sd = (real)sqrt(stats[2] / (cnt - 1.0));
    denom = sd + eps;
  }
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    real v = x[i];
    if (stats) v = (v - mean) / denom;
    if (clip > 0) v = min(max(v, -clip), clip);
    y[i] = v;
  }
}

// 'accumulate' mode: out[n,p] = sum_{t=a..b} f(adv[n,t]) (inclusive end) with
// f = optional normalise + clamp (temporal_correlated_agent.py:211-228).
template <typename real>
__global__ __launch_bounds__(256) void segment_accumulate_kernel(
    const real* __restrict__ adv, const int64_t* __restrict__ pairs, int P,
    real* __restrict__ out, int64_t N, int T, const double* __restrict__ stats,
    real eps, real clip) {
  real mean = 0, denom = 1;
  if (stats) {
    mean = (real)stats[1];
    denom = (real)sqrt(stats[2] / (stats[0] - 1.0)) + eps;
  }
  const int64_t total = N * (int64_t)P;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / P;
    const int p = (int)(i - n * P);
    const int a = (int)pairs[2 * p], b = (int)pairs[2 * p + 1];
    const real* row = adv + n * (int64_t)T;
    real acc = 0;
    for (int t = a; t <= b && t < T; ++t) {
      real v = row[t];
      if (stats) v = (v - mean) / denom;
      if (clip > 0) v = min(max(v, -clip), clip);
      acc += v;
    }
    out[i] = acc;
  }
}

// 'accumulated_rewards' mode (temporal_correlated_agent.py:288-319): one block
// per pair: acc[n] = sum gamma^t r, out = (acc - mean_n acc) / gamma^a.
template <typename real>
__global__ __launch_bounds__(256) void segment_accrew_kernel(
    const real* __restrict__ rewards, const int64_t* __restrict__ pairs, int P,
    real* __restrict__ out, int64_t N, int T, real gamma,
    const real* __restrict__ col_mean, int center) {
  __shared__ double s_red[4];
  const int p = blockIdx.x;
  const int a = (int)pairs[2 * p], b = (int)pairs[2 * p + 1];
  double loc = 0;
  for (int64_t n = threadIdx.x; n < N; n += 256) {
    const real* row = rewards + n * (int64_t)T;
    real acc = 0;
    for (int t = a; t < b; ++t) acc += row[t] * pow(gamma, real(t));
    out[n * P + p] = acc;
    loc += (double)acc;
  }
  if (!center) return;
  const real mean = col_mean ? col_mean[p]
                             : (real)(block_sum(loc, s_red) / (double)N);
  const real d = pow(gamma, real(a));
  for (int64_t n = threadIdx.x; n < N; n += 256)
    out[n * P + p] = (out[n * P + p] - mean) / d;
}

inline int gae_cu_count() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess)
      n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <typename real>
int gae_launch(const real* rewards, const real* values, const uint8_t* dones,
               const uint8_t* tl_dones, real* adv, real* ret,
               const int64_t* pairs, int P, real* seg_out, double* partials,
               int64_t N, int T, real gamma, real lam, int use_gae,
               hipStream_t stream) {
  TCE_CHECK_ARG(N > 0 && T > 0, "gae: N and T must be positive");
  TCE_CHECK_ARG(rewards && values && dones && tl_dones && adv && ret,
                "gae: null buffer");
  TCE_CHECK_ARG(P >= 0 && P <= 256, "gae: at most 256 pairs in the fused path");
  TCE_CHECK_ARG(P == 0 || (pairs && seg_out && partials),
                "gae: segment outputs missing");
  const int64_t nblocks = ceil_div(N, GAE_ENVS_PER_BLOCK);
  TCE_CHECK_ARG(nblocks < (1ll << 31), "gae: too many envs");
  // several waves per SIMD: prefetch the whole tile (see the kernel)
  static const int pf_env = getenv("TCE_GAE_PF") ? atoi(getenv("TCE_GAE_PF")) : 0;
  const bool deep = pf_env ? pf_env >= 8 : nblocks >= 2 * gae_cu_count();
#define GAE_GO(G, V, NPS, R)                                                  \
  do {                                                                        \
    if (deep)                                                                 \
      hipLaunchKernelGGL((gae_dpp_kernel<real, G, V, NPS, 8, R>),             \
                         dim3((unsigned)nblocks), dim3(GAE_BT), 0, stream,    \
                         rewards, values, dones, tl_dones, adv, ret, N, T,    \
                         gamma, lam);                                         \
    else                                                                      \
      hipLaunchKernelGGL((gae_dpp_kernel<real, G, V, NPS, 4, R>),             \
                         dim3((unsigned)nblocks), dim3(GAE_BT), 0, stream,    \
                         rewards, values, dones, tl_dones, adv, ret, N, T,    \
                         gamma, lam);                                         \
  } while (0)
  const bool vec = (T % 4 == 0);
  const bool rag = !vec && T >= 4;          // 16-byte accesses for rows of any length >= 4
  const int np = (int)ceil_div(T, GAE_PSTEPS);
  if (use_gae && vec && np <= GAE_MAXP) {
    switch (np) {
      case 1: GAE_GO(true, true, 1, false); break;
      case 2: GAE_GO(true, true, 2, false); break;
      case 3: GAE_GO(true, true, 3, false); break;
      case 4: GAE_GO(true, true, 4, false); break;
      case 5: GAE_GO(true, true, 5, false); break;
      case 6: GAE_GO(true, true, 6, false); break;
      case 7: GAE_GO(true, true, 7, false); break;
      default: GAE_GO(true, true, 8, false); break;
    }
  } else if (use_gae && rag && np <= GAE_MAXP) {
    switch (np) {
      case 1: GAE_GO(true, true, 1, true); break;
      case 2: GAE_GO(true, true, 2, true); break;
      case 3: GAE_GO(true, true, 3, true); break;
      case 4: GAE_GO(true, true, 4, true); break;
      case 5: GAE_GO(true, true, 5, true); break;
      case 6: GAE_GO(true, true, 6, true); break;
      case 7: GAE_GO(true, true, 7, true); break;
      default: GAE_GO(true, true, 8, true); break;
    }
  } else if (use_gae) {
    if (vec) GAE_GO(true, true, 0, false);
    else if (rag) GAE_GO(true, true, 0, true);
    else GAE_GO(true, false, 0, false);
  } else {
    if (vec) GAE_GO(false, true, 0, false);
    else if (rag) GAE_GO(false, true, 0, true);
    else GAE_GO(false, false, 0, false);
  }
#undef GAE_GO
  TCE_LAUNCH_CHECK();
  if (P > 0) {
    hipLaunchKernelGGL(segadv_vs_kernel<real>, dim3((unsigned)ceil_div(N, 4)),
                       dim3(256), 0, stream, rewards, values, pairs, P, seg_out,
                       partials, N, T, gamma);
    TCE_LAUNCH_CHECK();
  }
  return 0;
}

}  // namespace

extern "C" {

int64_t tce_gae_num_partials(int64_t N) { return ceil_div(N, 4); }

int tce_gae_f32(const float* rewards, const float* values, const uint8_t* dones,
                const uint8_t* tl_dones, float* adv, float* ret,
                const int64_t* pairs, int P, float* seg_out, double* partials,
                int64_t N, int T, float gamma, float lam, int use_gae,
                void* stream) {
  return gae_launch<float>(rewards, values, dones, tl_dones, adv, ret, pairs, P,
                           seg_out, partials, N, T, gamma, lam, use_gae,
                           (hipStream_t)stream);
}

int tce_gae_f64(const double* rewards, const double* values, const uint8_t* dones,
                const uint8_t* tl_dones, double* adv, double* ret,
                const int64_t* pairs, int P, double* seg_out, double* partials,
                int64_t N, int T, double gamma, double lam, int use_gae,
                void* stream) {
  return gae_launch<double>(rewards, values, dones, tl_dones, adv, ret, pairs, P,
                            seg_out, partials, N, T, gamma, lam, use_gae,
                            (hipStream_t)stream);
}

#define TCE_MOMENTS_BLOCKS 512

int64_t tce_moments_num_partials(void) { return TCE_MOMENTS_BLOCKS; }

#define DEFINE_MOMENTS(SFX, REAL)                                              \
  int tce_moments_partial_##SFX(const REAL* x, int64_t n, double* partials,    \
                                void* stream) {                                \
    TCE_CHECK_ARG(x && partials && n > 0, "moments: bad arguments");           \
    hipLaunchKernelGGL(moments_partial_kernel<REAL>, dim3(TCE_MOMENTS_BLOCKS), \
                       dim3(256), 0, (hipStream_t)stream, x, n, partials);     \
    TCE_LAUNCH_CHECK();                                                        \
    return 0;                                                                  \
  }                                                                            \
  int tce_normalize_##SFX(const REAL* x, REAL* y, int64_t n,                   \
                          const double* stats, REAL eps, REAL clip,            \
                          int single_std_one, void* stream) {                  \
    TCE_CHECK_ARG(x && y && n > 0, "normalize: bad arguments");                \
    const int64_t nb = tmin<int64_t>(ceil_div(n, 256), 2048);                   \
    hipLaunchKernelGGL(normalize_kernel<REAL>, dim3((unsigned)nb), dim3(256),  \
                       0, (hipStream_t)stream, x, y, n, stats, eps, clip,      \
                       single_std_one);                                        \
    TCE_LAUNCH_CHECK();                                                        \
    return 0;                                                                  \
  }                                                                            \
  int tce_segment_accumulate_##SFX(const REAL* adv, const int64_t* pairs,      \
                                   int P, REAL* out, int64_t N, int T,         \
                                   const double* stats, REAL eps, REAL clip,   \
                                   void* stream) {                             \
    TCE_CHECK_ARG(adv && pairs && out && N > 0 && P > 0 && T > 0,              \
                  "segment_accumulate: bad arguments");                        \
    const int64_t nb = tmin<int64_t>(ceil_div(N * P, 256), 2048);               \
    hipLaunchKernelGGL(segment_accumulate_kernel<REAL>, dim3((unsigned)nb),    \
                       dim3(256), 0, (hipStream_t)stream, adv, pairs, P, out,  \
                       N, T, stats, eps, clip);                                \
    TCE_LAUNCH_CHECK();                                                        \
    return 0;                                                                  \
  }                                                                            \
  int tce_segment_accrew_##SFX(const REAL* rewards, const int64_t* pairs,      \
                               int P, REAL* out, int64_t N, int T, REAL gamma, \
                               const REAL* col_mean, int center,               \
                               void* stream) {                                 \
    TCE_CHECK_ARG(rewards && pairs && out && N > 0 && P > 0 && T > 0,          \
                  "segment_accrew: bad arguments");                            \
    hipLaunchKernelGGL(segment_accrew_kernel<REAL>, dim3(P), dim3(256), 0,     \
                       (hipStream_t)stream, rewards, pairs, P, out, N, T,      \
                       gamma, col_mean, center);                               \
    TCE_LAUNCH_CHECK();                                                        \
    return 0;                                                                  \
  }

DEFINE_MOMENTS(f32, float)
DEFINE_MOMENTS(f64, double)

int tce_moments_finalize(const double* partials, int nparts, double* stats,
                         void* stream) {
  TCE_CHECK_ARG(partials && stats && nparts > 0, "moments_finalize: bad arguments");
  hipLaunchKernelGGL(moments_finalize_kernel, dim3(1), dim3(256), 0,
                     (hipStream_t)stream, partials, nparts, stats);
  TCE_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
