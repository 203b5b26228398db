// Which SIMD does each wave of a 512-thread, 160 KB-LDS workgroup land on?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512, 1) void k(int* out) {
  extern __shared__ char smem[];
  smem[threadIdx.x] = 1;
  if ((threadIdx.x & 63) == 0) {
    const unsigned simd = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);
    const unsigned wid = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);
    out[blockIdx.x * 8 + (threadIdx.x >> 6)] = (int)(simd | (wid << 8));
  }
}
int main() {
  int* d; hipMalloc(&d, 64 * 8 * sizeof(int));
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(k, dim3(64), dim3(512), 160 * 1024, 0, d);
  int h[64 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 12; ++b) {
    printf("wg %2d: simd of waves 0..7:", b);
    for (int w = 0; w < 8; ++w) printf(" %d", h[b * 8 + w] & 255);
    printf("   wave slots:");
    for (int w = 0; w < 8; ++w) printf(" %d", h[b * 8 + w] >> 8);
    printf("\n");
  }
  return 0;
}
