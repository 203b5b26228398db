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
