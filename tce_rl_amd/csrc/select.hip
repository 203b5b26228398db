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

// pass == PASSES: the final launch (grid 1)
template <typename real>
__global__ __launch_bounds__(256) void median_pass_kernel(const real* __restrict__ x, int64_t n,
                                                          int pass, unsigned* __restrict__ ghist,
                                                          double* __restrict__ out) {
  typedef KeyT<real> K;
  typedef typename K::key key;
  __shared__ unsigned sc[256];
  __shared__ int sel[2];
  __shared__ unsigned lh[256];
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
    if (t == 0) {
      out[0] = *nan_count ? __longlong_as_double(0x7FF8000000000000ll) : (double)K::back(prefix);
      *nan_count = 0;
    }
    for (int q = 0; q < K::PASSES; ++q) ghist[q * 256 + t] = 0;    // ready for the next call
    return;
  }
  lh[t] = 0;
  __syncthreads();
  const int shift = 8 * (K::PASSES - 1 - pass);
  unsigned nans = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + t; i < n; i += (int64_t)gridDim.x * 256) {
    const real v = x[i];
    const key kk = K::of(v);
    if (pass == 0 && v != v) ++nans;
    if ((kk & mask) == prefix) atomicAdd(&lh[(unsigned)(kk >> shift) & 255u], 1u);
  }
  __syncthreads();
  if (lh[t]) atomicAdd(&ghist[pass * 256 + t], lh[t]);
  if (nans) atomicAdd(nan_count, nans);
}

template <typename real>
int median_launch(const real* x, int64_t n, double* out, unsigned* ws, hipStream_t st) {
  typedef KeyT<real> K;
  const int64_t want = (n + 4095) / 4096;
  const unsigned grid = (unsigned)(want < 1 ? 1 : (want > 1024 ? 1024 : want));
  for (int p = 0; p <= K::PASSES; ++p) {
    hipLaunchKernelGGL(median_pass_kernel<real>, dim3(p == K::PASSES ? 1 : grid), dim3(256), 0,
                       st, x, n, p, ws, out);
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
  return median_launch<float>(x, n, out, ws, (hipStream_t)stream);
}

int tce_median_f64(const double* x, int64_t n, double* out, unsigned* ws, void* stream) {
  TCE_CHECK_ARG(x && out && ws && n > 0 && n < ((int64_t)1 << 31),
                "median: null buffer / element count outside [1, 2^31)");
  return median_launch<double>(x, n, out, ws, (hipStream_t)stream);
}

}  // extern "C"
