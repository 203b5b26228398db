// cycles per v_mfma_f64_16x16x4_f64 issued by one wave per SIMD with 1 / 2 / 4
// independent accumulation chains (operands in registers).
// hipcc --offload-arch=gfx950 -O3 scripts/ubench_mfma64.hip -o scripts/ubench_mfma64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NCH>
__global__ __launch_bounds__(256, 1) void k(double* out, long long* cyc, int iters) {
  d4 acc[NCH];
  for (int c = 0; c < NCH; ++c) acc[c] = (d4){0, 0, 0, 0};
  double a = threadIdx.x * 0.001, b = threadIdx.x * 0.002 + 1.0;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NCH>
void run(int waves) {
  double* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 8); hipMalloc(&cyc, 8);
  const int iters = 1000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NCH>, dim3(256), dim3(waves * 64), 0, 0, out, cyc, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NCH>, dim3(256), dim3(waves * 64), 0, 0, out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 16 * NCH;
  printf("chains %d waves/CU %d: %.1f cycles per MFMA per wave, %.1f TFLOP/s chip, clock %.2f GHz\n", NCH, waves, c / n,
         n * waves * 256 * 2048 / (ms * 1e-3) / 1e12, c / (ms * 1e-3) / 1e9);
}
int main() {
  run<1>(4); run<2>(4); run<4>(4); run<2>(8); run<4>(8);
  return 0;
}
