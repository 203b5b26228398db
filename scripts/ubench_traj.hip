// Store-pattern micro-benchmark for the trajectory kernel: N x T x 8 floats.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int N = 4096, T = 500, C = 8, NBG = 6;
// A: lane = t, two 16-B stores 32 B apart per lane (current layout), params via SGPR
__global__ __launch_bounds__(256) void kA(const float* __restrict__ B, const float* __restrict__ w, float* __restrict__ out, int epb) {
  const int t = blockIdx.x * 256 + threadIdx.x; const int tc = t < T ? t : T - 1;
  float row[4 + 2 * NBG];
  for (int i = 0; i < 4 + 2 * NBG; ++i) row[i] = B[tc * (4 + 2 * NBG) + i];
  for (int n = blockIdx.y * epb; n < (blockIdx.y + 1) * epb; ++n) {
    const float* wn = w + n * 32;
    float o[8];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      float p = row[0] * wn[24 + d] + row[1] * wn[28 + d], v = row[2] * wn[24 + d] + row[3] * wn[28 + d];
#pragma unroll
      for (int b = 0; b < NBG; ++b) { p += row[4 + b] * wn[d * NBG + b]; v += row[4 + NBG + b] * wn[d * NBG + b]; }
      o[d] = p; o[4 + d] = v;
    }
    if (t < T) { f4* dst = (f4*)(out + ((size_t)n * T + t) * C); dst[0] = (f4){o[0], o[1], o[2], o[3]}; dst[1] = (f4){o[4], o[5], o[6], o[7]}; }
  }
}
// B: lane = (t, half): one contiguous 16-B store per lane
__global__ __launch_bounds__(256) void kB(const float* __restrict__ B, const float* __restrict__ w, float* __restrict__ out, int epb) {
  const int i = blockIdx.x * 256 + threadIdx.x; const int t = i >> 1, h = i & 1; const int tc = t < T ? t : T - 1;
  float row[2 + NBG];
  row[0] = B[tc * (4 + 2 * NBG) + 2 * h]; row[1] = B[tc * (4 + 2 * NBG) + 2 * h + 1];
  for (int b = 0; b < NBG; ++b) row[2 + b] = B[tc * (4 + 2 * NBG) + 4 + h * NBG + b];
  for (int n = blockIdx.y * epb; n < (blockIdx.y + 1) * epb; ++n) {
    const float* wn = w + n * 32;
    float o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      float p = row[0] * wn[24 + d] + row[1] * wn[28 + d];
#pragma unroll
      for (int b = 0; b < NBG; ++b) p += row[2 + b] * wn[d * NBG + b];
      o[d] = p;
    }
    if (t < T) *(f4*)(out + ((size_t)n * T + t) * C + 4 * h) = (f4){o[0], o[1], o[2], o[3]};
  }
}
// C: pure contiguous fill (store roofline)
__global__ __launch_bounds__(256) void kC(float* __restrict__ out, size_t n4) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) ((f4*)out)[i] = (f4){1, 2, 3, 4};
}
int main() {
  float *B, *w, *out; hipMalloc(&B, T * 16 * 4); hipMalloc(&w, N * 32 * 4); hipMalloc(&out, (size_t)N * T * C * 4);
  hipMemset(B, 0, T * 16 * 4); hipMemset(w, 0, N * 32 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto fn) {
    fn(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 20; ++i) fn(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %.1f us  %.2f TB/s\n", name, ms * 50, (double)N * T * C * 4 / (ms / 20 * 1e-3) / 1e12);
  };
  for (int epb : {4, 16, 64}) {
    char nm[64]; snprintf(nm, 64, "A lane=t strided16B epb=%d", epb);
    run(nm, [&] { hipLaunchKernelGGL(kA, dim3(2, N / epb), dim3(256), 0, 0, B, w, out, epb); });
    snprintf(nm, 64, "B lane=(t,half) contiguous epb=%d", epb);
    run(nm, [&] { hipLaunchKernelGGL(kB, dim3(4, N / epb), dim3(256), 0, 0, B, w, out, epb); });
  }
  run("C pure fill 2048 blocks", [&] { hipLaunchKernelGGL(kC, dim3(2048), dim3(256), 0, 0, out, (size_t)N * T * C / 4); });
  return 0;
}
