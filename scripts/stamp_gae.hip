// Diagnostic build: where does a gae_dpp wave spend its cycles?
#define GAE_STAMP
#include "../tce_rl_amd/csrc/gae.hip"
#include <vector>
#include <algorithm>
extern "C" void tce_set_error(const char* m) { fprintf(stderr, "err: %s\n", m); }
int main() {
  const int64_t N = 4096; const int T = 500;
  float *r, *v, *adv, *ret; uint8_t *d, *tl;
  hipMalloc(&r, N * T * 4); hipMalloc(&v, N * (T + 1) * 4); hipMalloc(&adv, N * T * 4);
  hipMalloc(&ret, N * T * 4); hipMalloc(&d, N * T); hipMalloc(&tl, N * T);
  hipMemset(r, 0, N * T * 4); hipMemset(v, 0, N * (T + 1) * 4); hipMemset(d, 0, N * T); hipMemset(tl, 0, N * T);
  unsigned long long* st; hipMalloc(&st, 1024 * 8 * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &st, sizeof(st));
  for (int it = 0; it < 3; ++it)
    gae_launch<float>(r, v, d, tl, adv, ret, nullptr, 0, nullptr, nullptr, N, T, 1.0f, 0.95f, 1, 0);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(1024 * 8); hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  double ph[2] = {0}; unsigned long long mn = ~0ull, mx = 0;
  for (int b = 0; b < 1024; ++b) { ph[0] += double(h[b * 8 + 1] - h[b * 8]); ph[1] += double(h[b * 8 + 4] - h[b * 8 + 1]); mn = std::min(mn, h[b * 8]); mx = std::max(mx, h[b * 8 + 4]); }
  printf("avg cycles/wave: issue-loads %.0f  passes %.0f | first-start..last-end %llu cycles\n", ph[0] / 1024, ph[1] / 1024, mx - mn);
  return 0;
}
