// The exact 16x16x4 matrix instructions of gfx950 for float and double behind one
// name.  Operands: lane l holds A[l % 16][l / 16] and B[l / 16][l % 16]; result
// register i of lane l holds D[mfma16_row(l / 16, i)][l % 16] -- the row order of
// the two result layouts differs (scripts/probe_mfma_layout.hip).
#pragma once
#include <hip/hip_runtime.h>

typedef float mfma16_f32x4 __attribute__((ext_vector_type(4)));
typedef double mfma16_f64x4 __attribute__((ext_vector_type(4)));
template <typename real> struct Mfma16;
template <> struct Mfma16<float> { typedef mfma16_f32x4 acc; };
template <> struct Mfma16<double> { typedef mfma16_f64x4 acc; };

__device__ inline mfma16_f32x4 mfma16(float a, float b, mfma16_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ inline mfma16_f64x4 mfma16(double a, double b, mfma16_f64x4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
template <typename real> __device__ inline int mfma16_row(int g, int i) {
  return sizeof(real) == 4 ? 4 * g + i : 4 * i + g;
}
