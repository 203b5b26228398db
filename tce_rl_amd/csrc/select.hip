// Exact median of a device array by radix select: the statistic the reference's
// metric dictionaries carry for every dataset tensor of an iteration
// (util.generate_stats, mprl/util/util_numerical.py:130-164, called from
// temporal_correlated_agent.py:166-176 / black_box_agent.py:83-103).  A
// library median sorts the array; for the five 2 M-element tensors of a C2
// dataset those sorts were 0.8 ms of every step (they run on the policy's
// stream beside the persistent critic grid and keep it at 224 workgroups).
//
// Keys: the IEEE bits made monotone (sign flip for positives, complement for
// negatives), 8 bits per pass from the top.  Pass p histograms byte p of the
// keys that match the prefix chosen so far; every workgroup re-derives that
// prefix from the histograms of the earlier passes (a 256-bin scan each), so no
// state is passed between the launches except the histograms themselves.  The
// last launch (one workgroup) walks all histograms, writes the element of rank
// (n - 1) / 2 -- the lower median, torch.median's convention -- as a double and
// zeroes the workspace for the next call.  torch.median propagates NaN: pass 0
// counts the NaN inputs (either sign; as keys they would sort above +inf /
// below -inf) and the final launch writes NaN if there was one.
#include "common.h"

namespace {

template <typename real> struct KeyT;
template <> struct KeyT<float> {
  typedef uint32_t key;
  static constexpr int PASSES = 4;
  static __device__ inline key of(float v) {
    const uint32_t u = __float_as_uint(v);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
  }
  static __device__ inline float back(key k) {
    const uint32_t u = k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu);
    return __uint_as_float(u);
  }
};
template <> struct KeyT<double> {
  typedef uint64_t key;
  static constexpr int PASSES = 8;
  static __device__ inline key of(double v) {
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    return u ^ ((u >> 63) ? 0xFFFFFFFFFFFFFFFFull : 0x8000000000000000ull);
  }
  static __device__ inline double back(key k) {
    const uint64_t u = k ^ ((k >> 63) ? 0x8000000000000000ull : 0xFFFFFFFFFFFFFFFFull);
    return __longlong_as_double((long long)u);
  }
};

// The bucket of `hist` (256 bins, thread t holds bin t) that contains rank k;
// k becomes the rank inside that bucket.  All 256 threads call it.
__device__ inline int pick_bucket(unsigned h, int64_t& k, unsigned* sc, int* sel) {
  const int t = threadIdx.x;
  sc[t] = h;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const unsigned v = t >= off ? sc[t - off] : 0u;
    __syncthreads();
    sc[t] += v;
    __syncthreads();
  }
  const int64_t incl = sc[t], excl = incl - h;
  if (k >= excl && k < incl) { sel[0] = t; sel[1] = (int)excl; }
  __syncthreads();
  const int b = sel[0];
  k -= sel[1];
  __syncthreads();
  return b;
}

// STATS: the other four entries of generate_stats ride along -- pass 0 reads every
// element anyway and also sums d = x - x[0], d^2 (double) and takes max / min per
// workgroup (partials behind the histograms, fixed-order combination in the final
// launch): out[0..4] = {mean, max, min, median, std (n - 1)} instead of out[0] =
// median.  One chain of PASSES + 1 launches per tensor where the metric
// dictionaries took four library reductions, a stack and the select.
constexpr int ST_MAX_BLOCKS = 1024;
__device__ inline double* stats_partials(unsigned* ghist) {
  return reinterpret_cast<double*>(ghist + 8 * 256 + 2);
}

// pass == PASSES: the final launch (grid 1)
// The sums are taken about a representative value (cancellation in
// s2 - n md^2); an infinite or NaN first element is no representative -- the
// differences would all be inf - inf = NaN where torch returns +-inf.
__device__ inline double stats_shift(double x0) { return isfinite(x0) ? x0 : 0.0; }

template <typename real, bool STATS>
__global__ __launch_bounds__(256) void median_pass_kernel(const real* __restrict__ x, int64_t n,
                                                          int pass, unsigned* __restrict__ ghist,
                                                          double* __restrict__ out,
                                                          unsigned nblocks0) {
  typedef KeyT<real> K;
  typedef typename K::key key;
  __shared__ unsigned sc[256];
  __shared__ int sel[2];
  __shared__ unsigned lh[256];
  __shared__ double red[4];
  const int t = threadIdx.x;
  int64_t k = (n - 1) / 2;
  key prefix = 0, mask = 0;
  for (int q = 0; q < pass; ++q) {
    const int shift = 8 * (K::PASSES - 1 - q);
    const int b = pick_bucket(ghist[q * 256 + t], k, sc, sel);
    prefix |= (key)b << shift;
    mask |= (key)0xFF << shift;
  }
  unsigned* nan_count = ghist + 8 * 256;
  if (pass == K::PASSES) {
    const double med =
        *nan_count ? __longlong_as_double(0x7FF8000000000000ll) : (double)K::back(prefix);
    if (STATS) {
      // combine the workgroups' partials in workgroup order (wave 0; fixed order)
      if (t < 64) {
        const double* part = stats_partials(ghist);
        double s1 = 0, s2 = 0, mx = -INFINITY, mn = INFINITY;
        for (unsigned b = t; b < nblocks0; b += 64) {
          s1 += part[4 * b];
          s2 += part[4 * b + 1];
          mx = fmax(mx, part[4 * b + 2]);
          mn = fmin(mn, part[4 * b + 3]);
        }
        s1 = wave_sum_f64(s1);
        s2 = wave_sum_f64(s2);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          mx = fmax(mx, __shfl_xor(mx, off, 64));
          mn = fmin(mn, __shfl_xor(mn, off, 64));
        }
        if (t == 0) {
          const double dn = (double)n, shift = stats_shift((double)x[0]);
          const double md = s1 / dn;
          out[0] = shift + md;
          // (max / min of a tensor with a NaN are NaN in torch: the NaN count decides)
          const double nanv = __longlong_as_double(0x7FF8000000000000ll);
          out[1] = *nan_count ? nanv : mx;
          out[2] = *nan_count ? nanv : mn;
          out[3] = med;
          out[4] = n > 1 ? sqrt(fmax(s2 - dn * md * md, 0.0) / (dn - 1.0)) : 0.0;
          if (*nan_count) out[0] = out[4] = nanv;          // (fmax drops a NaN)
        }
      }
      __syncthreads();
      if (t == 0) *nan_count = 0;
    } else if (t == 0) {
      out[0] = med;
      *nan_count = 0;
    }
    for (int q = 0; q < K::PASSES; ++q) ghist[q * 256 + t] = 0;    // ready for the next call
    return;
  }
  lh[t] = 0;
  __syncthreads();
  const int shift = 8 * (K::PASSES - 1 - pass);
  unsigned nans = 0;
  const bool st = STATS && pass == 0;
  const double shift0 = st ? stats_shift((double)x[0]) : 0.0;
  double s1 = 0, s2 = 0, mx = -INFINITY, mn = INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * 256 + t; i < n; i += (int64_t)gridDim.x * 256) {
    const real v = x[i];
    const key kk = K::of(v);
    if (pass == 0 && v != v) ++nans;
    if ((kk & mask) == prefix) atomicAdd(&lh[(unsigned)(kk >> shift) & 255u], 1u);
    if (st) {
      const double d = (double)v - shift0;
      s1 += d;
      s2 += d * d;
      mx = fmax(mx, (double)v);
      mn = fmin(mn, (double)v);
    }
  }
  __syncthreads();
  if (lh[t]) atomicAdd(&ghist[pass * 256 + t], lh[t]);
  if (nans) atomicAdd(nan_count, nans);
  if (st) {
    s1 = wave_sum_f64(s1);
    s2 = wave_sum_f64(s2);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mx = fmax(mx, __shfl_xor(mx, off, 64));
      mn = fmin(mn, __shfl_xor(mn, off, 64));
    }
    __shared__ double wpart[4][4];
    if ((t & 63) == 0) {
      wpart[t >> 6][0] = s1; wpart[t >> 6][1] = s2; wpart[t >> 6][2] = mx; wpart[t >> 6][3] = mn;
    }
    __syncthreads();
    if (t == 0) {
      double* part = stats_partials(ghist) + 4 * (int64_t)blockIdx.x;
      part[0] = wpart[0][0] + wpart[1][0] + wpart[2][0] + wpart[3][0];
      part[1] = wpart[0][1] + wpart[1][1] + wpart[2][1] + wpart[3][1];
      part[2] = fmax(fmax(wpart[0][2], wpart[1][2]), fmax(wpart[2][2], wpart[3][2]));
      part[3] = fmin(fmin(wpart[0][3], wpart[1][3]), fmin(wpart[2][3], wpart[3][3]));
    }
  }
  (void)red;
}

template <typename real, bool STATS>
int median_launch(const real* x, int64_t n, double* out, unsigned* ws, hipStream_t st) {
  typedef KeyT<real> K;
  const int64_t want = (n + 4095) / 4096;
  const unsigned grid =
      (unsigned)(want < 1 ? 1 : (want > ST_MAX_BLOCKS ? ST_MAX_BLOCKS : want));
  for (int p = 0; p <= K::PASSES; ++p) {
    hipLaunchKernelGGL((median_pass_kernel<real, STATS>), dim3(p == K::PASSES ? 1 : grid),
                       dim3(256), 0, st, x, n, p, ws, out, grid);
    TCE_LAUNCH_CHECK();
  }
  return 0;
}

}  // namespace

extern "C" {

int tce_median_ws_len(void) { return 8 * 256 + 1; }

int tce_median_f32(const float* x, int64_t n, double* out, unsigned* ws, void* stream) {
  TCE_CHECK_ARG(x && out && ws && n > 0 && n < ((int64_t)1 << 31),
                "median: null buffer / element count outside [1, 2^31)");
  return median_launch<float, false>(x, n, out, ws, (hipStream_t)stream);
}

int tce_median_f64(const double* x, int64_t n, double* out, unsigned* ws, void* stream) {
  TCE_CHECK_ARG(x && out && ws && n > 0 && n < ((int64_t)1 << 31),
                "median: null buffer / element count outside [1, 2^31)");
  return median_launch<double, false>(x, n, out, ws, (hipStream_t)stream);
}

// out5 = {mean, max, min, median, std (n - 1)} of x [n] (doubles)
int tce_stats5_ws_len(void) { return 8 * 256 + 2 + 2 * 4 * ST_MAX_BLOCKS; }

int tce_stats5_f32(const float* x, int64_t n, double* out5, unsigned* ws, void* stream) {
  TCE_CHECK_ARG(x && out5 && ws && n > 0 && n < ((int64_t)1 << 31),
                "stats5: null buffer / element count outside [1, 2^31)");
  return median_launch<float, true>(x, n, out5, ws, (hipStream_t)stream);
}

int tce_stats5_f64(const double* x, int64_t n, double* out5, unsigned* ws, void* stream) {
  TCE_CHECK_ARG(x && out5 && ws && n > 0 && n < ((int64_t)1 << 31),
                "stats5: null buffer / element count outside [1, 2^31)");
  return median_launch<double, true>(x, n, out5, ws, (hipStream_t)stream);
}

}  // extern "C"
