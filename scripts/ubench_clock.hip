// What clock does a short, latency-bound kernel actually get?  s_memtime
// (shader cycles) vs s_memrealtime (100 MHz) around a dependent chain.
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang fp contract(off)
__global__ void k(float* out, unsigned long long* o, int steps, float a, float kk) {
  float x = out[threadIdx.x];
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int c = 0; c < steps / 16; ++c) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { float kx = kk * x; x = a + kx; }
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) { o[0] = c1 - c0; o[1] = r1 - r0; }
}
int main() {
  float* out; unsigned long long* o; hipMalloc(&out, 4096); hipMalloc(&o, 16); hipMemset(out, 0, 4096);
  unsigned long long h[2];
  int cfgs[][2] = {{1, 512}, {1, 8192}, {1024, 512}, {1024, 8192}, {1024, 262144}};
  for (auto& c : cfgs) for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(c[0]), dim3(64), 0, 0, out, o, c[1], 0.01f, 0.95f); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
    printf("blocks %4d steps %6d: %.1f cyc/step, clock %.0f MHz, event %.1f us\n", c[0], c[1], double(h[0]) / c[1], double(h[0]) / (double(h[1]) / 100.0), ms * 1000);
  }
  return 0;
}
