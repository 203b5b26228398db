// Which physical CUs does a CU-masked stream use?  For a few masks, launch a
// grid of workgroups and histogram (xcc, se, cu) of where they ran.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <vector>
#include <set>
#include <map>
__global__ void k(unsigned* out) {
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);    // HW_ID[15:0]
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);   // XCC_ID[3:0]
    out[blockIdx.x] = hw | (xcc << 16);
  }
  // keep the CU busy a little so that workgroups spread
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(8);
}
static void run(const std::vector<uint32_t>& mask, const char* name) {
  hipStream_t st;
  if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
  unsigned* d; hipMalloc(&d, 4096 * 4);
  hipLaunchKernelGGL(k, dim3(2048), dim3(64), 0, st, d);
  hipStreamSynchronize(st);
  std::vector<unsigned> h(2048); hipMemcpy(h.data(), d, 2048 * 4, hipMemcpyDeviceToHost);
  std::map<unsigned, std::set<unsigned>> per_xcc;
  for (unsigned v : h) per_xcc[v >> 16].insert(((v >> 13) & 7) * 100 + ((v >> 12) & 1) * 50 + ((v >> 8) & 15));
  printf("%s:", name);
  for (auto& kv : per_xcc) {
    printf("  xcc%u[%zu]:", kv.first, kv.second.size());
    int n = 0; for (unsigned c : kv.second) if (n++ < 6) printf(" %u", c);
  }
  printf("\n");
  hipFree(d); hipStreamDestroy(st);
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); printf("CUs %d\n", p.multiProcessorCount);
  std::vector<uint32_t> all(8, 0xffffffffu); run(all, "all 256 bits");
  for (int b : {0, 1, 2, 7, 8, 9, 31, 32, 33, 64, 255}) {
    std::vector<uint32_t> m(8, 0); m[b / 32] = 1u << (b % 32);
    char nm[32]; sprintf(nm, "bit %d", b); run(m, nm);
  }
  { std::vector<uint32_t> m(8, 0); m[0] = 0xffffffffu; run(m, "bits 0-31"); }
  { std::vector<uint32_t> m(8, 0); m[7] = 0xffffffffu; run(m, "bits 224-255"); }
  { std::vector<uint32_t> m(8, 0x11111111u); run(m, "every 4th bit"); }
  return 0;
}
