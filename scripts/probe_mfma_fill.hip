// How many independent VALU instructions ride for free between two bf16 MFMAs of one wave
// per SIMD?  Cycles per MFMA of a loop of v_mfma_f32_16x16x32_bf16 (one accumulator chain)
// with F v_add_f32 on unrelated registers after each, F = 0 .. 8, and the same with the
// 32x32x16 form.  build: hipcc --offload-arch=gfx950 -O3 scripts/probe_mfma_fill.hip -o
// scripts/probe_mfma_fill ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int F, bool BIG>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters) {
  bf8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
  f32x4 c = {0, 0, 0, 0};
  f32x16 d;
  for (int i = 0; i < 16; ++i) d[i] = 0.f;
  float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (BIG) d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d, 0, 0, 0);
      else c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#pragma unroll
      for (int f = 0; f < F; ++f) {
        if ((f & 3) == 0) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v0));
        if ((f & 3) == 1) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v1));
        if ((f & 3) == 2) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v2));
        if ((f & 3) == 3) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v3));
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = c[0] + d[0] + v0 + v1 + v2 + v3;
}
template <int F, bool BIG> void run(float* out, long long* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<F, BIG>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%s F=%d: %.2f cycles per MFMA\n", BIG ? "32x32x16" : "16x16x32", F, (double)h / (iters * 16.0));
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  run<0, false>(out, cyc); run<1, false>(out, cyc); run<2, false>(out, cyc); run<3, false>(out, cyc);
  run<4, false>(out, cyc); run<6, false>(out, cyc); run<8, false>(out, cyc);
  run<0, true>(out, cyc); run<2, true>(out, cyc); run<4, true>(out, cyc); run<6, true>(out, cyc);
  run<8, true>(out, cyc); run<12, true>(out, cyc);
  return 0;
}
