// Micro-benchmark: cycles per step of the serial recurrence x = a + k*x on one
// wave (4 active lanes), in registers and out of LDS.
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang fp contract(off)
constexpr int T = 512, CH = 16;
__global__ void k_reg(float* out, unsigned long long* cyc, float a, float k) {
  float x = out[threadIdx.x];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int c = 0; c < T / CH; ++c) {
#pragma unroll
    for (int i = 0; i < CH; ++i) { float kx = k * x; x = a + kx; }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE>
__global__ void k_lds(float* out, unsigned long long* cyc) {
  __shared__ float sa[4][T + 1], sk[4][T + 1], sx[4][T + 1];
  for (int i = threadIdx.x; i < 4 * T; i += blockDim.x) { sa[i / T][i % T] = 0.01f; sk[i / T][i % T] = 0.95f; }
  __syncthreads();
  if (threadIdx.x < 4) {
    float x = 0;
    float* pa = sa[threadIdx.x]; float* pk = sk[threadIdx.x]; float* px = MODE == 2 ? sx[threadIdx.x] : sa[threadIdx.x];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int j = T; j > 0; j -= CH) {
      float ra[CH], rk[CH], rx[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) { ra[i] = pa[j - 1 - i]; rk[i] = pk[j - 1 - i]; }
#pragma unroll
      for (int i = 0; i < CH; ++i) { float kx = rk[i] * x; x = ra[i] + kx; rx[i] = x; if (MODE == 0) px[j - 1 - i] = x; }
      if (MODE >= 1) {
#pragma unroll
        for (int i = 0; i < CH; ++i) px[j - 1 - i] = rx[i];
      }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
  }
}
template <int CHN>
__global__ void k_lds_db(float* out, unsigned long long* cyc) {
  __shared__ float sa[4][T + 1], sk[4][T + 1];
  for (int i = threadIdx.x; i < 4 * T; i += blockDim.x) { sa[i / T][i % T] = 0.01f; sk[i / T][i % T] = 0.95f; }
  __syncthreads();
  if (threadIdx.x < 4) {
    float x = 0;
    float* pa = sa[threadIdx.x]; float* pk = sk[threadIdx.x];
    float a0[CHN], k0[CHN], a1[CHN], k1[CHN];
    auto load = [&](float* ra, float* rk, int top) {
#pragma unroll
      for (int i = 0; i < CHN; ++i) { ra[i] = pa[top - 1 - i]; rk[i] = pk[top - 1 - i]; }
    };
    auto chain = [&](const float* ra, const float* rk, int top) {
#pragma unroll
      for (int i = 0; i < CHN; ++i) { float kx = rk[i] * x; x = ra[i] + kx; pa[top - 1 - i] = x; }
    };
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int j = T;
    load(a0, k0, j);
#pragma unroll 1
    for (;;) {
      if (j > CHN) load(a1, k1, j - CHN);
      chain(a0, k0, j); j -= CHN; if (j == 0) break;
      if (j > CHN) load(a0, k0, j - CHN);
      chain(a1, k1, j); j -= CHN; if (j == 0) break;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
  }
}
// software-pipelined: while chaining chunk c, store chunk c-1's results and
// load chunk c+1's inputs, one LDS op per step, all independent of the chain.
template <int CHN>
__global__ void k_lds_sp(float* out, unsigned long long* cyc) {
  __shared__ float sa[4][T + 1], sk[4][T + 1];
  for (int i = threadIdx.x; i < 4 * T; i += blockDim.x) { sa[i / T][i % T] = 0.01f; sk[i / T][i % T] = 0.95f; }
  __syncthreads();
  if (threadIdx.x < 4) {
    float x = 0;
    float* pa = sa[threadIdx.x]; float* pk = sk[threadIdx.x];
    float a0[CHN], k0[CHN], a1[CHN], k1[CHN], r0[CHN], r1[CHN];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int j = T;
#pragma unroll
    for (int i = 0; i < CHN; ++i) { a0[i] = pa[j - 1 - i]; k0[i] = pk[j - 1 - i]; r1[i] = 0; }
    // step(cur inputs, cur results, next inputs, prev results)
#define SP_CHUNK(A, K, R, AN, KN, RP)                                          \
    {                                                                           \
      const bool more = j > CHN;                                                \
      const bool prev = j < T;                                                  \
      _Pragma("unroll") for (int i = 0; i < CHN; ++i) {                         \
        float kx = K[i] * x; x = A[i] + kx; R[i] = x;                           \
        if (prev) pa[j + CHN - 1 - i] = RP[i];                                  \
        if (more) { AN[i] = pa[j - CHN - 1 - i]; KN[i] = pk[j - CHN - 1 - i]; } \
      }                                                                         \
      j -= CHN;                                                                 \
    }
#pragma unroll 1
    while (j > 0) {
      SP_CHUNK(a0, k0, r0, a1, k1, r1)
      if (j == 0) { _Pragma("unroll") for (int i = 0; i < CHN; ++i) pa[CHN - 1 - i] = r0[i]; break; }
      SP_CHUNK(a1, k1, r1, a0, k0, r0)
      if (j == 0) { _Pragma("unroll") for (int i = 0; i < CHN; ++i) pa[CHN - 1 - i] = r1[i]; break; }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x + pa[5];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
  }
}
int main() {
  float* out; unsigned long long* cyc; hipMalloc(&out, 1024); hipMalloc(&cyc, 8); hipMemset(out, 0, 1024);
  unsigned long long h;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_reg, dim3(1), dim3(64), 0, 0, out, cyc, 0.01f, 0.95f); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("reg chain: %.1f cyc/step\n", double(h) / T);
    hipLaunchKernelGGL(k_lds<0>, dim3(1), dim3(256), 0, 0, out, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("lds chain, in-place interleaved writes: %.1f cyc/step\n", double(h) / T);
    hipLaunchKernelGGL(k_lds<1>, dim3(1), dim3(256), 0, 0, out, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("lds chain, in-place batched writes: %.1f cyc/step\n", double(h) / T);
    hipLaunchKernelGGL(k_lds<2>, dim3(1), dim3(256), 0, 0, out, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("lds chain, separate output array, batched: %.1f cyc/step\n", double(h) / T);
    hipLaunchKernelGGL(k_lds_db<16>, dim3(1), dim3(256), 0, 0, out, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("lds chain, double-buffered CH16: %.1f cyc/step\n", double(h) / T);
    hipLaunchKernelGGL(k_lds_sp<16>, dim3(1), dim3(256), 0, 0, out, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("lds chain, software-pipelined CH16: %.1f cyc/step\n", double(h) / T);
    hipLaunchKernelGGL(k_lds_sp<8>, dim3(1), dim3(256), 0, 0, out, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("lds chain, software-pipelined CH8: %.1f cyc/step\n", double(h) / T);
    hipLaunchKernelGGL(k_lds_db<32>, dim3(1), dim3(256), 0, 0, out, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("lds chain, double-buffered CH32: %.1f cyc/step\n", double(h) / T);
    hipLaunchKernelGGL(k_lds_db<8>, dim3(1), dim3(256), 0, 0, out, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("lds chain, double-buffered CH8: %.1f cyc/step\n", double(h) / T);
  }
  return 0;
}
