// Cycle counts of the small-matrix primitives of csrc/klproj2.h (one workgroup):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I tce_rl_amd/csrc scripts/ubench_klp2.hip -o scripts/ubench_klp2
#include "klproj2.h"
#include <stdio.h>
#include <vector>
using namespace klp2;
__global__ __launch_bounds__(256) void bench(double* g, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* B0 = reinterpret_cast<double*>(smem_raw);
  double* B1 = B0 + SZ;
  double* B2 = B1 + SZ;
  double* strips = B2 + SZ;
  const Lane ln;
  load(B0, g, 64, false, true);
  each(B2, [&](int r, int c) { return B0[r * P + c]; });
  long long t0 = __builtin_readcyclecounter();
  const double ld = gj_inverse(B0, strips);
  long long t1 = __builtin_readcyclecounter();
  mm_nt(B1, B0, B0, ln, Ident());
  long long t2 = __builtin_readcyclecounter();
  mm_tn(B1, B0, B0, ln, Ident());
  long long t3 = __builtin_readcyclecounter();
  cholesky(B2, strips);
  long long t4 = __builtin_readcyclecounter();
  each(B1, [&](int r, int c) { return B0[r * P + c] + B2[c * P + r]; });
  long long t5 = __builtin_readcyclecounter();
  store_ctx(g, B1, 64);
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; cyc[4] = t5 - t4; g[4096] = ld; }
}
int main() {
  std::vector<double> h(4097, 0.0);
  for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) h[i * 64 + j] = (i == j ? 2.0 : 0.0) + 0.01 * ((i * 7 + j * 3) % 11) + 0.01 * ((j * 7 + i * 3) % 11);
  double* d; long long* c;
  hipMalloc(&d, 4097 * 8); hipMalloc(&c, 64);
  hipFuncSetAttribute((const void*)bench, hipFuncAttributeMaxDynamicSharedMemorySize, (3 * SZ + 1024) * 8);
  long long hc[5];
  for (int rep = 0; rep < 3; ++rep) {
    hipMemcpy(d, h.data(), 4097 * 8, hipMemcpyHostToDevice);
    bench<<<1, 256, (3 * SZ + 1024) * 8>>>(d, c);
    hipMemcpy(hc, c, 40, hipMemcpyDeviceToHost);
    printf("gj %lld  mm_nt %lld  mm_tn %lld  chol %lld  each %lld cycles\n", hc[0], hc[1], hc[2], hc[3], hc[4]);
  }
  return 0;
}
