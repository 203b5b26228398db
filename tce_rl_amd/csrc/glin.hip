// Generic dense layer on the exact 16x16x4 matrix instructions of gfx950
// (v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64) for the MLP shapes none of
// the fused families covers (csrc/mlp.hip 128 x 2, mlpw 256 x 2 / fp64, smlp
// 32 / 64 x 2, pmlp 128 x 1 / 128 x 2 / 256 x 1): the reference sizes its nets
// from arbitrary YAML numbers (mprl/util/util_hyperparams.py:8-46) and the
// contextual covariance head (mprl/rl/policy/abstract_policy.py:96-109) is a
// second MLP with K (K + 1) / 2 outputs -- until round 5 those ran on library
// GEMMs under torch autograd.
//
//   forward   y  [R][O] = x [R][I] W^T + b            MLP.forward, util_nn.py:225-246
//   backward  dx [R][I] = dy [R][O] W
//             dW [O][I] = dy^T x   (contracts over ALL rows: split over the rows,
//                                   partial tiles summed in a fixed order)
//             db [O]    = sum_r dy
//
// One kernel, three operand layouts: C[m][n] = sum_k A(m, k) B(n, k) with A / B
// addressed through two strides each; a 64 x 64 tile of C per workgroup, four
// waves of 16 rows x 64 columns, 16 k per LDS stage, the next stage's global
// loads in flight while the current one feeds the matrix cores.  These are the
// small, skinny products of policy-sized nets (a few thousand rows, widths up
// to a few hundred): latency-bound, not a roofline item -- what matters is that
// the arithmetic is the library's (fp32 / fp64 products, fp32 / fp64 accumulate)
// and that nothing leaves the hand-written path.
#include "common.h"
#include "mfma16.h"

extern "C" int64_t tce_sum_dim0_slices(int64_t N, int64_t M);
extern "C" int tce_sum_dim0_f32(const float* x, float* out, float* ws, int64_t N, int64_t M,
                                void* stream);
extern "C" int tce_sum_dim0_f64(const double* x, double* out, double* ws, int64_t N, int64_t M,
                                void* stream);

namespace {

constexpr int GL_BM = 64, GL_BN = 64, GL_BK = 16, GL_BT = 256;
constexpr int GL_P = GL_BK + 1;                 // odd pitch: fragment reads hit distinct banks
constexpr int GL_EPT = GL_BM * GL_BK / GL_BT;   // elements per thread and operand and stage (4)

template <typename real>
struct GlArgs {
  const real* a;            // A(m, k) = a[m * lda_m + k * lda_k]
  int64_t lda_m, lda_k;
  const real* b;            // B(n, k) = b[n * ldb_n + k * ldb_k]
  int64_t ldb_n, ldb_k;
  real* c;                  // C[m * ldc + n]
  int64_t ldc;
  const real* bias;         // [N], nullable (ignored when the k range is split)
  int64_t M, K;
  int N;
  int64_t k_per_split;      // a multiple of GL_BK; gridDim.z splits
  real* partial;            // [gridDim.z][M][N] when gridDim.z > 1
};

// M_CONTIG / N_CONTIG: which index of the operand is the unit-stride one (decides
// the thread -> element map of the stage so that neighbouring lanes read
// neighbouring addresses)
template <typename real, bool A_MCONTIG, bool B_NCONTIG>
__global__ __launch_bounds__(GL_BT) void glin_kernel(GlArgs<real> g) {
  __shared__ real As[GL_BM * GL_P];
  __shared__ real Bs[GL_BN * GL_P];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * GL_BM;     // (x: up to 2^31 - 1 row tiles)
  const int n0 = (int)blockIdx.y * GL_BN;
  const int64_t k_lo = (int64_t)blockIdx.z * g.k_per_split;
  const int64_t k_hi = tmin<int64_t>(g.K, k_lo + g.k_per_split);
  typename Mfma16<real>::acc acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = typename Mfma16<real>::acc{0, 0, 0, 0};

  real ra[GL_EPT], rb[GL_EPT];
  auto fetch = [&](int64_t k0) {
#pragma unroll
    for (int j = 0; j < GL_EPT; ++j) {
      const int e = tid + GL_BT * j;
      {
        const int r = A_MCONTIG ? (e & (GL_BM - 1)) : (e / GL_BK);
        const int kk = A_MCONTIG ? (e / GL_BM) : (e & (GL_BK - 1));
        const int64_t m = m0 + r, k = k0 + kk;
        const bool ok = m < g.M && k < k_hi;
        ra[j] = ok ? g.a[m * g.lda_m + k * g.lda_k] : real(0);
      }
      {
        const int r = B_NCONTIG ? (e & (GL_BN - 1)) : (e / GL_BK);
        const int kk = B_NCONTIG ? (e / GL_BN) : (e & (GL_BK - 1));
        const int64_t n = n0 + r, k = k0 + kk;
        const bool ok = n < g.N && k < k_hi;
        rb[j] = ok ? g.b[n * g.ldb_n + k * g.ldb_k] : real(0);
      }
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int j = 0; j < GL_EPT; ++j) {
      const int e = tid + GL_BT * j;
      {
        const int r = A_MCONTIG ? (e & (GL_BM - 1)) : (e / GL_BK);
        const int kk = A_MCONTIG ? (e / GL_BM) : (e & (GL_BK - 1));
        As[r * GL_P + kk] = ra[j];
      }
      {
        const int r = B_NCONTIG ? (e & (GL_BN - 1)) : (e / GL_BK);
        const int kk = B_NCONTIG ? (e / GL_BN) : (e & (GL_BK - 1));
        Bs[r * GL_P + kk] = rb[j];
      }
    }
  };
  if (k_lo < k_hi) fetch(k_lo);
  const int fr = lane & 15, fk = lane >> 4;
  for (int64_t k0 = k_lo; k0 < k_hi; k0 += GL_BK) {
    __syncthreads();                       // the previous stage has been consumed
    stash();
    __syncthreads();
    if (k0 + GL_BK < k_hi) fetch(k0 + GL_BK);   // in flight during the MFMAs below
#pragma unroll
    for (int ks = 0; ks < GL_BK; ks += 4) {
      const real av = As[(16 * wave + fr) * GL_P + ks + fk];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const real bv = Bs[(16 * t + fr) * GL_P + ks + fk];
        acc[t] = mfma16(av, bv, acc[t]);
      }
    }
  }
  const bool split = gridDim.z > 1;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int n = n0 + 16 * t + fr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = m0 + 16 * wave + mfma16_row<real>(fk, i);
      if (m < g.M && n < g.N) {
        if (split) {
          g.partial[((int64_t)blockIdx.z * g.M + m) * g.N + n] = acc[t][i];
        } else {
          g.c[m * g.ldc + n] = acc[t][i] + (g.bias ? g.bias[n] : real(0));
        }
      }
    }
  }
}

// c[m][n] = sum over the splits, in split order
template <typename real>
__global__ __launch_bounds__(256) void glin_reduce_kernel(const real* __restrict__ partial,
                                                          int splits, int64_t MN, int N,
                                                          real* __restrict__ c, int64_t ldc) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= MN) return;
  real s = 0;
  for (int z = 0; z < splits; ++z) s += partial[(int64_t)z * MN + e];
  const int64_t m = e / N;
  c[m * ldc + (e - m * N)] = s;
}

inline int gl_cu_count() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess)
      n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

// splits of the contraction for dW (K = rows): enough workgroups for the chip,
// never shorter than 256 rows each
inline int gl_splits(int64_t M, int N, int64_t K) {
  const int64_t tiles = ceil_div(M, GL_BM) * ceil_div(N, GL_BN);
  int64_t s = ceil_div(2 * gl_cu_count(), tiles);
  s = tmin<int64_t>(s, ceil_div(K, 256));
  return (int)tmax<int64_t>(1, tmin<int64_t>(s, 1024));
}

template <typename real, bool AM, bool BN>
int gl_launch(GlArgs<real> g, int splits, hipStream_t st) {
  TCE_CHECK_ARG(g.M > 0 && g.N > 0 && g.K > 0, "glin: empty product");
  TCE_CHECK_ARG(ceil_div(g.M, GL_BM) < (1ll << 31), "glin: too many rows");
  g.k_per_split = ceil_div(ceil_div(g.K, splits), GL_BK) * GL_BK;
  splits = (int)ceil_div(g.K, g.k_per_split);
  TCE_CHECK_ARG(splits == 1 || g.partial, "glin: split-K workspace missing");
  const dim3 grid((unsigned)ceil_div(g.M, GL_BM), (unsigned)ceil_div(g.N, GL_BN), (unsigned)splits);
  hipLaunchKernelGGL((glin_kernel<real, AM, BN>), grid, dim3(GL_BT), 0, st, g);
  TCE_LAUNCH_CHECK();
  if (splits > 1) {
    const int64_t MN = g.M * g.N;
    hipLaunchKernelGGL(glin_reduce_kernel<real>, dim3((unsigned)ceil_div(MN, 256)), dim3(256), 0,
                       st, (const real*)g.partial, splits, MN, g.N, g.c, g.ldc);
    TCE_LAUNCH_CHECK();
  }
  return 0;
}

inline int gl_sum_dim0(const float* x, float* out, float* ws, int64_t N, int64_t M, void* st) {
  return tce_sum_dim0_f32(x, out, ws, N, M, st);
}
inline int gl_sum_dim0(const double* x, double* out, double* ws, int64_t N, int64_t M, void* st) {
  return tce_sum_dim0_f64(x, out, ws, N, M, st);
}

constexpr int GL_MAX_DIM = 4096;

template <typename real>
int gl_forward(const real* x, int64_t x_stride, int64_t R, int din, int dout, const real* W,
               const real* bias, real* y, void* stream) {
  TCE_CHECK_ARG(x && W && y && R > 0, "glin_forward: null buffer / no rows");
  TCE_CHECK_ARG(din >= 1 && din <= GL_MAX_DIM && dout >= 1 && dout <= GL_MAX_DIM,
                "glin_forward: 1 <= D_in, D_out <= 4096");
  TCE_CHECK_ARG(x_stride >= din, "glin_forward: row stride below D_in");
  GlArgs<real> g{x, x_stride, 1, W, din, 1, y, dout, bias, R, din, dout, 0, nullptr};
  return gl_launch<real, false, false>(g, 1, (hipStream_t)stream);
}

template <typename real>
int64_t gl_ws_len(int64_t R, int din, int dout) {
  const int64_t part = (int64_t)gl_splits(dout, din, R) * dout * din;
  const int64_t sums = tce_sum_dim0_slices(R, dout) * dout;
  return part + sums + 8;
}

template <typename real>
int gl_backward(const real* x, int64_t x_stride, const real* gy, const real* W, int64_t R,
                int din, int dout, real* gx, real* gW, real* gb, real* ws, void* stream) {
  TCE_CHECK_ARG(x && gy && W && gW && ws && R > 0, "glin_backward: null buffer / no rows");
  TCE_CHECK_ARG(din >= 1 && din <= GL_MAX_DIM && dout >= 1 && dout <= GL_MAX_DIM,
                "glin_backward: 1 <= D_in, D_out <= 4096");
  hipStream_t st = (hipStream_t)stream;
  if (gx) {
    // dx[r][i] = sum_o dy[r][o] W[o][i]:  A = dy (k = o contiguous), B(n = i, k = o) = W[o][i]
    GlArgs<real> g{gy, dout, 1, W, 1, din, gx, din, nullptr, R, dout, din, 0, nullptr};
    const int rc = gl_launch<real, false, true>(g, 1, st);
    if (rc) return rc;
  }
  {
    // dW[o][i] = sum_r dy[r][o] x[r][i]:  A(m = o, k = r) = dy[r][o], B(n = i, k = r) = x[r][i]
    GlArgs<real> g{gy, 1, dout, x, 1, x_stride, gW, din, nullptr, dout, R, din, 0, ws};
    const int rc = gl_launch<real, true, true>(g, gl_splits(dout, din, R), st);
    if (rc) return rc;
  }
  if (gb) {
    real* sws = ws + (int64_t)gl_splits(dout, din, R) * dout * din;
    return gl_sum_dim0(gy, gb, sws, R, dout, stream);
  }
  return 0;
}

}  // namespace

extern "C" {

int tce_glin_max_dim(void) { return GL_MAX_DIM; }

int64_t tce_glin_ws_len(int64_t R, int din, int dout) { return gl_ws_len<float>(R, din, dout); }

int tce_glin_forward_f32(const float* x, int64_t x_stride, int64_t R, int din, int dout,
                         const float* W, const float* bias, float* y, void* stream) {
  return gl_forward<float>(x, x_stride, R, din, dout, W, bias, y, stream);
}
int tce_glin_forward_f64(const double* x, int64_t x_stride, int64_t R, int din, int dout,
                         const double* W, const double* bias, double* y, void* stream) {
  return gl_forward<double>(x, x_stride, R, din, dout, W, bias, y, stream);
}
int tce_glin_backward_f32(const float* x, int64_t x_stride, const float* grad_y, const float* W,
                          int64_t R, int din, int dout, float* grad_x, float* grad_W,
                          float* grad_b, float* ws, void* stream) {
  return gl_backward<float>(x, x_stride, grad_y, W, R, din, dout, grad_x, grad_W, grad_b, ws,
                            stream);
}
int tce_glin_backward_f64(const double* x, int64_t x_stride, const double* grad_y,
                          const double* W, int64_t R, int din, int dout, double* grad_x,
                          double* grad_W, double* grad_b, double* ws, void* stream) {
  return gl_backward<double>(x, x_stride, grad_y, W, R, din, dout, grad_x, grad_W, grad_b, ws,
                             stream);
}

}  // extern "C"
