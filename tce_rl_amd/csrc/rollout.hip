// Rollout-buffer kernels for gfx950: observation running mean/std (k12) and
// the non-MDP -> MDP reward re-shaping (k13).
//
//   RunningMeanStd.update / update_from_moments   mprl/util/util_numerical.py:315-337
//   TemporalCorrelatedSampler.apply_normalization mprl/rl/sampler/temporal_correlated_sampler.py:87-89
//   make_mdp_reward                               mprl/util/util_experiment.py:261-328
//
// col_moments: one pass over x [R, D] (R = N*(T+1) rows, HBM-bound: R*D*4 B):
// thread <-> (row lane, column) so that a workgroup reads whole rows
// (coalesced), per-thread shifted sums in double, LDS column reduction, one
// partial per workgroup; rms_finalize merges them into the running statistics
// with the reference's parallel-moments formula (unbiased batch variance).
#include "common.h"

namespace {

constexpr int RM_BT = 256;

// V = elements per thread (4 when D % 4 == 0: 16-byte loads, else 1).
template <typename real, int V>
__global__ __launch_bounds__(RM_BT) void col_moments_kernel(
    const real* __restrict__ x, int64_t R, int D, const real* __restrict__ shift,
    double* __restrict__ partials /* [gridDim.x][D][2] */) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  typedef real vec __attribute__((ext_vector_type(V), aligned(V * sizeof(real))));
  const int DV = D / V;                                   // column groups per row
  const int rpp = RM_BT / DV;                             // row lanes per pass
  double* s1 = reinterpret_cast<double*>(smem_raw);       // [rpp][D]
  double* s2 = s1 + rpp * D;
  const int tid = threadIdx.x;
  const int rl = tid / DV, cg = tid - rl * DV;
  const bool live = rl < rpp;
  double k[V], a1[V], a2[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    k[i] = (live && shift) ? (double)shift[cg * V + i] : 0.0;
    a1[i] = 0;
    a2[i] = 0;
  }
  if (live) {
    const int64_t per = (R + gridDim.x - 1) / gridDim.x;
    const int64_t lo = blockIdx.x * per, hi = tmin<int64_t>(R, lo + per);
    int64_t r = lo + rl;
    // 4 rows in flight per thread (independent loads, clamped addresses)
    for (; r < hi; r += 4 * rpp) {
      vec v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t rr = tmin<int64_t>(r + u * rpp, hi - 1);
        v[u] = *reinterpret_cast<const vec*>(x + rr * D + cg * V);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (r + u * rpp < hi) {
#pragma unroll
          for (int i = 0; i < V; ++i) {
            const double d = (double)v[u][i] - k[i];
            a1[i] += d;
            a2[i] += d * d;
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < V; ++i) {
      s1[rl * D + cg * V + i] = a1[i];
      s2[rl * D + cg * V + i] = a2[i];
    }
  }
  __syncthreads();
  if (tid < D) {
    double t1 = 0, t2 = 0;
    for (int i = 0; i < rpp; ++i) { t1 += s1[i * D + tid]; t2 += s2[i * D + tid]; }
    partials[((int64_t)blockIdx.x * D + tid) * 2 + 0] = t1;
    partials[((int64_t)blockIdx.x * D + tid) * 2 + 1] = t2;
  }
}
template <typename real>
__device__ inline real vec_get(real v, int) { return v; }

// One workgroup per column: sum the partials, then merge into the running
// statistics (update_from_moments, util_numerical.py:321-337).
template <typename real>
__global__ __launch_bounds__(256) void rms_finalize_kernel(
    const double* __restrict__ partials, int nparts, int D,
    const real* shift /* may alias mean */, double batch_count, double count,
    real* mean, real* var) {
  __shared__ double red[4];
  const int c = blockIdx.x;
  double t1 = 0, t2 = 0;
  for (int i = threadIdx.x; i < nparts; i += 256) {
    t1 += partials[((int64_t)i * D + c) * 2 + 0];
    t2 += partials[((int64_t)i * D + c) * 2 + 1];
  }
  t1 = block_sum(t1, red);
  t2 = block_sum(t2, red);
  if (threadIdx.x != 0) return;
  const double k = shift ? (double)shift[c] : 0.0;
  const double n = batch_count;
  const double b_mean = k + t1 / n;
  const double b_var = n > 1 ? (t2 - t1 * t1 / n) / (n - 1.0) : (double)NAN;  // unbiased
  const double m = (double)mean[c], v = (double)var[c];
  const double delta = b_mean - m;
  const double tot = count + n;
  const double new_mean = m + delta * n / tot;
  const double m2 = v * count + b_var * n + delta * delta * count * n / tot;
  mean[c] = (real)new_mean;
  var[c] = (real)(m2 / tot);
}

// y = (x - mean) / sqrt(var + eps), column = flat index % D
template <typename real>
__global__ __launch_bounds__(256) void rms_normalize_kernel(
    const real* __restrict__ x, real* __restrict__ y, int64_t total, int D,
    const real* __restrict__ mean, const real* __restrict__ var, real eps) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % D);
    y[i] = (x[i] - mean[c]) / sqrt(var[c] + eps);
  }
}

// make_mdp_reward: one thread per env row.
template <typename real>
__global__ __launch_bounds__(256) void mdp_reward_kernel(real* __restrict__ r,
                                                         const uint8_t* __restrict__ flags,
                                                         int64_t N, int T) {
  const int64_t n = blockIdx.x * 256ll + threadIdx.x;
  if (n >= N) return;
  real* row = r + n * (int64_t)T;
  const uint8_t* f = flags + n * (int64_t)T;
  int first = 0;                     // argmax of a 0/1 row: first 1, else 0
  bool found = false;
  real after = 0;
  for (int t = 0; t < T; ++t) {
    if (f[t]) {
      if (!found) { first = t; found = true; }
      after += row[t];
    }
  }
  if (first > 0) {                   // reference: event_index_first > 0
    row[first] = after;
    for (int t = first + 1; t < T; ++t) row[t] = 0;
  }
}

}  // namespace

// A keyed pseudo-random permutation of [0, n) WITHOUT a sort (agent option
// minibatch_permutation: device -- the critic's minibatch permutations drawn on
// the GPU instead of by numpy's sequential Fisher-Yates on the host,
// mprl/util/util_data_structure.py:378-391): out[i] = pi(i), pi = a balanced
// Feistel network on 2 h bits (2^(2h) >= n the smallest such power of four)
// restricted to [0, n) by cycle walking -- every element is computed on its own,
// O(1) expected rounds (the domain is < 4 n), no library call, 8 B written per
// row.  (Bijective-shuffle construction; not uniform over all n! permutations,
// as no 64-bit-keyed generator is -- what minibatching needs is that every row
// lands in an unpredictable piece, which tests/test_minibatch_gpu.py checks.)
__device__ inline unsigned feistel_round(unsigned r, unsigned k) {
  unsigned x = (r ^ k) * 0x9E3779B1u;
  x ^= x >> 15;
  x *= 0x85EBCA77u;
  x ^= x >> 13;
  x *= 0xC2B2AE3Du;
  x ^= x >> 16;
  return x;
}
__global__ __launch_bounds__(256) void feistel_permutation_kernel(int64_t* __restrict__ out,
                                                                  int64_t n, int h,
                                                                  unsigned long long key) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const unsigned long long mask = (1ull << h) - 1ull;
  unsigned keys[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
    keys[q] = feistel_round((unsigned)(key >> (8 * (q & 3))) + 0x632BE5ABu * (unsigned)q,
                            (unsigned)(key >> 32) ^ (0x7F4A7C15u * (unsigned)(q + 1)));
  unsigned long long x = (unsigned long long)i;
  do {
    unsigned long long l = x >> h, r = x & mask;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const unsigned long long f = (unsigned long long)feistel_round((unsigned)r, keys[q]) & mask;
      const unsigned long long nl = r;
      r = l ^ f;
      l = nl;
    }
    x = (l << h) | r;
  } while (x >= (unsigned long long)n);        // cycle walking: stays a bijection on [0, n)
  out[i] = (int64_t)x;
}

// select_batch (mprl/util/util_data_structure.py:362-375) for the row kernels of
// the black-box critic: x_out[i, :] = x[idx[i], :din], a_out[i] = a[idx[i]],
// b_out[i] = b[idx[i]] (b nullable) -- one launch per minibatch; the envs of a
// black-box batch are a few MB, so gathered copies cost nothing (the big TCE
// critics read the index inside their epoch kernels instead: csrc/mlp.hip).
template <typename real>
__global__ __launch_bounds__(256) void gather_rows_kernel(
    const real* __restrict__ x, int64_t x_stride, const real* __restrict__ a,
    const real* __restrict__ b, const int64_t* __restrict__ idx, int64_t n, int din,
    real* __restrict__ x_out, real* __restrict__ a_out, real* __restrict__ b_out) {
  const int64_t total = n * din;
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t i = e / din;
    const int f = (int)(e - i * din);
    x_out[e] = x[idx[i] * x_stride + f];
  }
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = idx[i];
    a_out[i] = a[r];
    if (b) b_out[i] = b[r];
  }
}

extern "C" {

int tce_feistel_permutation(int64_t* out, int64_t n, uint64_t key, void* stream) {
  TCE_CHECK_ARG(out && n > 0 && n < (1ll << 62), "feistel_permutation: bad arguments");
  int h = 1;
  while (h < 31 && (1ll << (2 * h)) < n) ++h;
  TCE_CHECK_ARG((1ll << (2 * h)) >= n, "feistel_permutation: n above 2^62");
  hipLaunchKernelGGL(feistel_permutation_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, out, n, h, (unsigned long long)key);
  TCE_LAUNCH_CHECK();
  return 0;
}

#define TCE_RMS_BLOCKS 1024

int64_t tce_rms_num_partials(void) { return TCE_RMS_BLOCKS; }

#define DEFINE_ROLLOUT(SFX, REAL)                                                  \
  int tce_rms_update_##SFX(const REAL* x, int64_t R, int D, REAL* mean, REAL* var, \
                           double count, double* partials_ws, void* stream) {      \
    TCE_CHECK_ARG(x && mean && var && partials_ws && R > 0 && D > 0 && D <= 256,   \
                  "rms_update: bad arguments (D <= 256)");                         \
    /* the running mean is the shift of the one-pass sums; it must not change   \
       while the partial kernel reads it: finalize runs after it on the stream */ \
    const bool v4 = (D % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0);    \
    const int rpp = RM_BT / (v4 ? D / 4 : D);                                      \
    const size_t lds = 2 * (size_t)rpp * D * sizeof(double);                       \
    const int nb = (int)tmin<int64_t>(TCE_RMS_BLOCKS, ceil_div(R, rpp));           \
    if (v4)                                                                        \
      hipLaunchKernelGGL((col_moments_kernel<REAL, 4>), dim3(nb), dim3(RM_BT),     \
                         lds, (hipStream_t)stream, x, R, D, mean, partials_ws);    \
    else                                                                           \
      hipLaunchKernelGGL((col_moments_kernel<REAL, 1>), dim3(nb), dim3(RM_BT),     \
                         lds, (hipStream_t)stream, x, R, D, mean, partials_ws);    \
    TCE_LAUNCH_CHECK();                                                            \
    hipLaunchKernelGGL(rms_finalize_kernel<REAL>, dim3(D), dim3(256), 0,           \
                       (hipStream_t)stream, partials_ws, nb, D, mean, (double)R,   \
                       count, mean, var);                                          \
    TCE_LAUNCH_CHECK();                                                            \
    return 0;                                                                      \
  }                                                                                \
  int tce_rms_normalize_##SFX(const REAL* x, REAL* y, int64_t total, int D,        \
                              const REAL* mean, const REAL* var, REAL eps,         \
                              void* stream) {                                      \
    TCE_CHECK_ARG(x && y && mean && var && total > 0 && D > 0,                     \
                  "rms_normalize: bad arguments");                                 \
    const int64_t nb = tmin<int64_t>(ceil_div(total, 256), 8192);                  \
    hipLaunchKernelGGL(rms_normalize_kernel<REAL>, dim3((unsigned)nb), dim3(256),  \
                       0, (hipStream_t)stream, x, y, total, D, mean, var, eps);    \
    TCE_LAUNCH_CHECK();                                                            \
    return 0;                                                                      \
  }                                                                                \
  int tce_gather_rows_##SFX(const REAL* x, int64_t x_stride, const REAL* a,        \
                            const REAL* b, const int64_t* idx, int64_t n, int din, \
                            REAL* x_out, REAL* a_out, REAL* b_out, void* stream) { \
    TCE_CHECK_ARG(x && a && idx && x_out && a_out && (!b || b_out) && n > 0 &&     \
                      din > 0,                                                     \
                  "gather_rows: bad arguments");                                   \
    const int64_t nb = tmin<int64_t>(ceil_div(n * din, 256), 4096);                \
    hipLaunchKernelGGL(gather_rows_kernel<REAL>, dim3((unsigned)nb), dim3(256), 0, \
                       (hipStream_t)stream, x, x_stride, a, b, idx, n, din, x_out, \
                       a_out, b_out);                                              \
    TCE_LAUNCH_CHECK();                                                            \
    return 0;                                                                      \
  }                                                                                \
  int tce_mdp_reward_##SFX(REAL* rewards, const uint8_t* event_flags, int64_t N,   \
                           int T, void* stream) {                                  \
    TCE_CHECK_ARG(rewards && event_flags && N > 0 && T > 0,                        \
                  "mdp_reward: bad arguments");                                    \
    hipLaunchKernelGGL(mdp_reward_kernel<REAL>, dim3((unsigned)ceil_div(N, 256)),  \
                       dim3(256), 0, (hipStream_t)stream, rewards, event_flags, N, \
                       T);                                                         \
    TCE_LAUNCH_CHECK();                                                            \
    return 0;                                                                      \
  }

DEFINE_ROLLOUT(f32, float)
DEFINE_ROLLOUT(f64, double)

}  // extern "C"
