// Probe of the operand / result lane layouts of the two exact 16x16x4 matrix
// instructions used by the wide-MLP kernels (f32 and f64): A = one-hot,
// B = coded values -> the result tells which (row, col) each lane register holds.
//   hipcc --offload-arch=gfx950 -O2 scripts/probe_mfma_layout.hip -o scripts/probe_mfma_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef double d4 __attribute__((ext_vector_type(4)));

// A[m][k] = (m+1) * 100 at k == kk only; B[k][n] = n + 1 at k == kk only
// assuming the f32 layouts: a lane l holds A[l%16][l/16], B[l/16][l%16].
template <typename T, typename V>
__global__ void probe(T* out, int kk, int mode) {
  const int l = threadIdx.x;
  T a = 0, b = 0;
  if (mode == 0) {          // assumed layout
    a = (l / 16 == kk) ? T((l % 16 + 1) * 100) : T(0);
    b = (l / 16 == kk) ? T(l % 16 + 1) : T(0);
  } else {                  // identify: a = lane id coded, b = 1 on all
    a = T(l + 1);
    b = (l == kk) ? T(1) : T(0);   // single B lane set
  }
  V c = {0, 0, 0, 0};
  if constexpr (sizeof(T) == 4)
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  else
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}

template <typename T, typename V>
void run(const char* name) {
  T* d;
  hipMalloc(&d, 256 * sizeof(T));
  T h[256];
  printf("== %s, mode 0 (A[m][k]=100(m+1), B[k][n]=n+1 at k=1): expect D[m][n]=100(m+1)(n+1)\n", name);
  probe<T, V><<<1, 64>>>(d, 1, 0);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int i = 0; i < 4; ++i) {
      const int m = 4 * (l / 16) + i, n = l % 16;
      if (h[l * 4 + i] != T(100 * (m + 1) * (n + 1))) ++bad;
    }
  printf("   f32-style D layout (lane l, reg i) = D[4*(l/16)+i][l%%16]: %s (%d mismatches)\n",
         bad ? "NO" : "yes", bad);
  if (bad) {
    for (int l = 0; l < 64; l += 5)
      printf("   lane %2d: %g %g %g %g\n", l, (double)h[l * 4], (double)h[l * 4 + 1],
             (double)h[l * 4 + 2], (double)h[l * 4 + 3]);
    // mode 1: B one-hot at lane kk: D[m][n(kk)] = A[m][k(kk)] = lane id of the A holder + 1
    for (int kk : {0, 1, 16, 17, 33, 50}) {
      probe<T, V><<<1, 64>>>(d, kk, 1);
      hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      printf("   B one-hot at lane %d -> nonzero results:", kk);
      for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i)
          if (h[l * 4 + i] != 0) printf(" (l%d,r%d)=%g", l, i, (double)h[l * 4 + i]);
      printf("\n");
    }
  }
  hipFree(d);
}

int main() {
  run<float, f4>("v_mfma_f32_16x16x4_f32");
  run<double, d4>("v_mfma_f64_16x16x4_f64");
  return 0;
}
