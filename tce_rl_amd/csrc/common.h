// Shared helpers for the gfx950 kernels of libtce_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

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
