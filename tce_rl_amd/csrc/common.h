// Shared helpers for the gfx950 kernels of libtce_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <unordered_map>

#define TCE_WAVE 64

extern "C" void tce_set_error(const char* msg);

#define TCE_CHECK_ARG(cond, msg)                                        \
  do {                                                                  \
    if (!(cond)) {                                                      \
      tce_set_error(msg);                                               \
      return 1;                                                         \
    }                                                                   \
  } while (0)

#define TCE_LAUNCH_CHECK()                                              \
  do {                                                                  \
    hipError_t e__ = hipGetLastError();                                 \
    if (e__ != hipSuccess) {                                            \
      tce_set_error(hipGetErrorString(e__));                            \
      return 2;                                                         \
    }                                                                   \
  } while (0)

// Raise a kernel's dynamic-LDS limit (needed above 48 KiB) ONCE per kernel and
// size, not on every launch: the attribute call is a runtime round trip on the
// launch path.
inline void tce_lds_limit(const void* kern, size_t bytes) {
  static std::unordered_map<const void*, size_t> have;
  if (bytes <= 48 * 1024) return;
  size_t& h = have[kern];
  if (bytes > h) {
    (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    h = bytes;
  }
}

template <typename T> __host__ __device__ inline T tmin(T a, T b) { return a < b ? a : b; }
template <typename T> __host__ __device__ inline T tmax(T a, T b) { return a > b ? a : b; }

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Sum over the 64 lanes of a wave (all lanes get the result).
template <typename T>
__device__ inline T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, TCE_WAVE);
  return v;
}

// ---- cross-lane sums of doubles on the VALU (DPP / permlane swaps) instead of
// LDS-routed shuffles: a double __shfl_xor is two ds_bpermute round trips.
template <int CTRL>
__device__ inline double dpp_perm_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// sum over each aligned group of 8 lanes (all 8 get it)
__device__ inline double dpp_sum8(double v) {
  v += dpp_perm_f64<0x141>(v);          // row_half_mirror: lane i <-> 7 - i
  v += dpp_perm_f64<0x4E>(v);           // quad_perm [2,3,0,1]
  v += dpp_perm_f64<0xB1>(v);           // quad_perm [1,0,3,2]
  return v;
}
// sum over the 64 lanes of a wave (all lanes get it)
__device__ inline double wave_sum_f64(double v) {
  v = dpp_sum8(v);
  v += dpp_perm_f64<0x140>(v);          // row_mirror: the other 8-group of the row
  {                                     // rows l ^ 16, then l ^ 32 (gfx950 swaps)
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
  }
  {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    v = __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
  }
  return v;
}

// Block-wide sum through LDS scratch (>= blockDim/64 entries); result valid in
// every thread.  Deterministic order.
template <typename T>
__device__ inline T block_sum(T v, T* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  T s = 0;
  for (int i = 0; i < nw; ++i) s += scratch[i];
  return s;
}

// ---------------------------------------------------------------------------
// One element of torch.optim.Adam with L2-in-gradient weight decay
// (mprl/rl/agent/abstract_agent.py:62-82), written so that EVERY kernel that
// applies it -- the stand-alone steps of csrc/optim.hip, the finish kernels of
// the fused epochs, the exchange kernels of csrc/xchg.hip -- produces the same
// bits from the same inputs: contraction is off and each product-sum is an
// explicit fma, so neither the vectorizer nor the surrounding code decides how
// the expression rounds (the fused and the stand-alone paths are compared bit
// for bit in tests/).  g: the gradient with its clip / shard factor applied.
template <typename real>
__device__ __forceinline__ void adam_coef(real lr, real b1, real b2, real step, real& step_size,
                                          real& bc2s) {
#pragma clang fp contract(off)
  const real bc1 = real(1) - pow(b1, step);
  bc2s = sqrt(real(1) - pow(b2, step));
  step_size = lr / bc1;
}
template <typename real>
__device__ __forceinline__ void adam_elem(real g, real& w, real& m, real& v, real b1, real b2,
                                          real eps, real wd, real step_size, real bc2s) {
#pragma clang fp contract(off)
  if (wd != real(0)) g = fma(wd, w, g);
  const real g1 = (real(1) - b1) * g;
  const real g2 = ((real(1) - b2) * g) * g;
  m = fma(b1, m, g1);
  v = fma(b2, v, g2);
  const real den = sqrt(v) / bc2s + eps;
  w = w - (step_size * m) / den;
}
