// Policy mean net on rows: D_in -> H (-> H) -> K with N rows (one per env), the
// networks the TCE / BBRL policies build for every task
// (mprl/rl/policy/abstract_policy.py:58-99 -> mprl/util/util_nn.py:75-246), in
// float32 AND float64, one or two hidden layers, H in {128, 256}, K <= 64:
//   box pushing   float64, 128 x 2 leaky_relu
//                 (mprl/config/box_push_random_init/tcp/entire/shared.yaml:7,75-78)
//   table tennis  float32, 256 x 1 tanh
//                 (mprl/config/table_tennis_4d/tcp/entire/shared.yaml:78-81)
// These ran on library GEMMs + autograd until round 3 (73 % of the launches of a
// box-pushing step).  Two launches per epoch now:
//   pmlp_fwd_kernel  x -> h1 (-> h2) -> mean, activations kept for the backward
//   pmlp_bwd_kernel  dL/dmean -> every weight / bias gradient (per-workgroup
//                    slabs), + pmlp_reduce_kernel (fixed-order slab sum).
//
// Mapping.  A workgroup (4 waves) walks over tiles of 32 rows.  Every product
// is a chain of exact 16x16x4 matrix instructions (v_mfma_f32_16x16x4_f32 /
// v_mfma_f64_16x16x4_f64: the fp64 form has the fp64 VALU's rate, but one
// instruction does 1024 FMAs on two 8-byte operands per lane, so LDS and issue
// slots stay free).  Operand fragments come from LDS images with row pitch
// P = 2 (mod 32) elements, read in two ways that are both conflict free:
//   "row" reads   lane (x = l % 16, q = l / 16) -> image[x][k + q]
//                 (A[m = row][k] of a forward / input-gradient product)
//   "unit" reads  lane -> image[s + 8 q][u0 + x], s = 0..7
//                 (A / B of a weight gradient, contraction over the 32 rows taken
//                  in the order s + 8 q: (8 q P) mod 32 = 16 q puts the two
//                  16-lane halves of an LDS lane group on disjoint banks)
// Weights stream from L2 through LDS in chunks of 32 contraction steps
// (forward: [unit][32], pitch 34; backward W used as stored, [32 rows][H],
// pitch H + 16), fetched into registers while the previous chunk is multiplied.
// The weight gradients of a workgroup stay in its accumulators over all its
// tiles (wave w owns the blocks whose column block = w mod 4).
#include "common.h"
#include "../../include/tce_hip.h"

extern "C" int tce_cu_budget_value(void);       // csrc/pair_logprob.hip

namespace {

enum { PM_TANH = 0, PM_RELU = 1, PM_LEAKY = 2, PM_SOFTPLUS = 3 };
constexpr int PM_RT = 32;            // rows per tile
constexpr int PM_BT = 256;           // threads per workgroup (4 waves)
constexpr int PM_KC = 32;            // contraction steps per weight chunk
constexpr int PM_WP = PM_KC + 2;     // pitch of a forward weight chunk

template <typename real> struct PT;
template <> struct PT<float> {
  typedef float acc __attribute__((ext_vector_type(4)));
  typedef float v2 __attribute__((ext_vector_type(2), aligned(8)));
};
template <> struct PT<double> {
  typedef double acc __attribute__((ext_vector_type(4)));
  typedef double v2 __attribute__((ext_vector_type(2), aligned(16)));
};
__device__ inline PT<float>::acc pmma(float a, float b, PT<float>::acc c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ inline PT<double>::acc pmma(double a, double b, PT<double>::acc c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
// tile row held by register i of lane group g (scripts/probe_mfma_layout.hip)
template <typename real> __device__ inline int pm_drow(int g, int i) {
  return sizeof(real) == 4 ? 4 * g + i : 4 * i + g;
}
__host__ __device__ constexpr int pm_pitch(int n) { return n <= 2 ? 2 : ((n - 2 + 31) / 32) * 32 + 2; }
__host__ __device__ constexpr int pm_up(int n, int m) { return (n + m - 1) / m * m; }

template <typename real> __device__ inline real pm_act(int act, real y) {
  switch (act) {
    case PM_TANH: return tanh(y);
    case PM_RELU: return y > real(0) ? y : real(0);
    case PM_LEAKY: return y > real(0) ? y : real(0.01) * y;
    default: return y > real(20) ? y : log1p(exp(y));
  }
}
// derivative expressed with the OUTPUT h = act(y)
template <typename real> __device__ inline real pm_act_d(int act, real h) {
  switch (act) {
    case PM_TANH: return real(1) - h * h;
    case PM_RELU: return h > real(0) ? real(1) : real(0);
    case PM_LEAKY: return h > real(0) ? real(1) : real(0.01);
    default: return h > real(20) ? real(1) : -expm1(-h);
  }
}

template <typename real> struct PmLayer {
  const real *W, *b;        // [Dout][Din], [Dout]
  real* hout;               // [N][H] activations kept for the backward (hidden layers; nullable)
  int Din, Dout;
};
template <typename real> struct PmArgs {
  const real* x;            // [N] rows of din features, stride x_stride
  int64_t x_stride, N;
  int din, K, act;
  const real* param;        // W1 [H][din] | b1 | (W2 [H][H] | b2) | W3 [K][H] | b3
  real *h1, *h2, *out;      // forward outputs (h2: two hidden layers)
  const real* g;            // backward: dL/dout [N][K]
  real* partials;           // backward: [gridDim.x][P]
  PmLayer<real> lay[3];     // forward: the layers in order (read by a runtime index: scalar loads)
};

template <int H, int NL> __host__ __device__ inline int64_t pm_num_params(int din, int K) {
  return (int64_t)H * din + H + (NL == 2 ? (int64_t)H * H + H : 0) + (int64_t)K * H + K;
}

// ---- forward ---------------------------------------------------------------------
template <typename real, int H>
__host__ __device__ inline size_t pm_fwd_lds(int din) {
  return sizeof(real) * ((size_t)PM_RT * pm_pitch(pm_up(din, 4)) + (size_t)PM_RT * pm_pitch(H) +
                         (size_t)H * PM_WP);
}

// (two workgroups per CU: while one waits for a weight chunk the other multiplies)
// (its LDS -- 153 KB -- allows one workgroup per CU only for float64 x 256)
template <typename real, int H, int NL>
__global__ __launch_bounds__(PM_BT, (sizeof(real) * H > 1024 ? 1 : 2)) void pmlp_fwd_kernel(
    PmArgs<real> a) {
  typedef typename PT<real>::acc acc_t;
  typedef typename PT<real>::v2 v2;
  constexpr int NBH = H / 64;                   // unit blocks per wave of a hidden layer
  constexpr int NPW = H * (PM_KC / 2) / PM_BT;  // element pairs per thread of a weight chunk
  constexpr int PH = pm_pitch(H);
  extern __shared__ __attribute__((aligned(16))) char pm_smem[];
  const int din = a.din, K = a.K, act = a.act;
  const int din4 = pm_up(din, 4), PX = pm_pitch(din4);
  real* xbuf = reinterpret_cast<real*>(pm_smem);
  real* hbuf = xbuf + PM_RT * PX;
  real* wbuf = hbuf + PM_RT * PH;
  const int t = threadIdx.x, l = t & 63, w = t >> 6, l16 = l & 15, q = l >> 4;
  const int64_t ntiles = (a.N + PM_RT - 1) / PM_RT;
  v2 regs[NPW];
  // chunk ch of layer ly -> registers (pairs; zero outside the matrix)
  auto fetch = [&](int ly, int ch) {
    const real* W = a.lay[ly].W;
    const int Di = a.lay[ly].Din, Do = a.lay[ly].Dout, k0 = ch * PM_KC;
    const bool vec = (Di & 1) == 0;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int p = t + PM_BT * i, u = p >> 4, k = k0 + 2 * (p & 15);
      v2 v = {real(0), real(0)};
      if (u < Do && k < Di) {
        const real* src = W + (u * Di + k);
        if (vec) v = *reinterpret_cast<const v2*>(src);
        else { v[0] = src[0]; if (k + 1 < Di) v[1] = src[1]; }
      }
      regs[i] = v;
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int p = t + PM_BT * i, u = p >> 4, kk = 2 * (p & 15);
      *reinterpret_cast<v2*>(wbuf + u * PM_WP + kk) = regs[i];
    }
  };
  fetch(0, 0);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t r0 = tile * PM_RT;
    // x tile (zero filled past din / N); the barriers of the first chunk order it
    for (int e = t; e < PM_RT * PX; e += PM_BT) {
      const int r = e / PX, c = e - r * PX;
      real v = 0;
      if (r0 + r < a.N && c < din) v = a.x[(r0 + r) * a.x_stride + c];
      xbuf[e] = v;
    }
#pragma unroll 1
    for (int ly = 0; ly <= NL; ++ly) {
      const real* in = ly == 0 ? xbuf : hbuf;
      const int Pin = ly == 0 ? PX : PH;
      const int Di = a.lay[ly].Din, Do = a.lay[ly].Dout;
      const int Di4 = pm_up(Di, 4);
      const int nub = (Do + 15) / 16;                 // unit blocks of this layer
      const int nch = (Di + PM_KC - 1) / PM_KC;
      acc_t c[2][NBH];
#pragma unroll
      for (int j = 0; j < NBH; ++j) c[0][j] = c[1][j] = acc_t{0, 0, 0, 0};
      real bias[NBH];
#pragma unroll
      for (int j = 0; j < NBH; ++j) bias[j] = 0;
#pragma unroll 1
      for (int ch = 0; ch < nch; ++ch) {
        __syncthreads();                              // the chunk buffer is free
        stash();
        __syncthreads();
        // the next chunk (of this layer, the next layer, or the next tile's first)
        if (ch + 1 < nch) fetch(ly, ch + 1);
        else fetch(ly < NL ? ly + 1 : 0, 0);
        if (ch == nch - 1) {
          const real* bp = a.lay[ly].b;
#pragma unroll
          for (int j = 0; j < NBH; ++j) {
            const int u = (w + 4 * j) * 16 + l16;
            bias[j] = u < Do ? bp[u] : real(0);
          }
        }
        const int k0 = ch * PM_KC, ns = (tmin(PM_KC, Di4 - k0)) / 4;
        const real* ap = in + l16 * Pin + k0 + q;
        const real* bp2 = wbuf + (w * 16 + l16) * PM_WP + q;
#pragma unroll 2
        for (int ks = 0; ks < ns; ++ks) {
          const real a0 = ap[ks * 4], a1 = ap[16 * Pin + ks * 4];
#pragma unroll
          for (int j = 0; j < NBH; ++j) {
            if (w + 4 * j < nub) {                    // uniform per wave
              const real b = bp2[j * 64 * PM_WP + ks * 4];
              c[0][j] = pmma(a0, b, c[0][j]);
              c[1][j] = pmma(a1, b, c[1][j]);
            }
          }
        }
      }
      // epilogue: every wave is done reading the input image of this layer
      __syncthreads();
      real* hout = a.lay[ly].hout;
#pragma unroll
      for (int j = 0; j < NBH; ++j) {
        if (w + 4 * j >= nub) continue;
        const int u = (w + 4 * j) * 16 + l16;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = rb * 16 + pm_drow<real>(q, i);
            real v = c[rb][j][i] + bias[j];
            if (ly < NL) {
              v = pm_act<real>(act, v);
              hbuf[r * PH + u] = v;
              if (hout && r0 + r < a.N) hout[(r0 + r) * H + u] = v;
            } else if (u < K && r0 + r < a.N) {
              a.out[(r0 + r) * K + u] = v;
            }
          }
      }
    }
  }
}

// ---- backward --------------------------------------------------------------------
template <typename real, int H> struct PmB {
  // rows of W (as stored, [rows][H]) per chunk of an input-gradient product
  static constexpr int KC2 = sizeof(real) * H > 1024 ? 16 : 32;
  static constexpr int PW2 = H + 16;
};
template <typename real, int H, int NL, int DB>
__host__ __device__ inline size_t pm_bwd_lds() {
  return sizeof(real) * ((size_t)PM_RT * pm_pitch(64) + (size_t)NL * PM_RT * pm_pitch(H) +
                         (size_t)PM_RT * pm_pitch(16 * DB) +
                         (size_t)PmB<real, H>::KC2 * PmB<real, H>::PW2);
}

template <typename real, int H, int NL, int DB>
__global__ __launch_bounds__(PM_BT) void pmlp_bwd_kernel(PmArgs<real> a) {
  typedef typename PT<real>::acc acc_t;
  typedef typename PT<real>::v2 v2;
  constexpr int NBH = H / 64, HB = H / 16;
  constexpr int KC2 = PmB<real, H>::KC2, PW2 = PmB<real, H>::PW2;
  constexpr int NPW = KC2 * (H / 2) / PM_BT;     // pairs per thread of a weight chunk
  constexpr int PH = pm_pitch(H), PG = pm_pitch(64), PX = pm_pitch(16 * DB);
  extern __shared__ __attribute__((aligned(16))) char pm_smem[];
  const int din = a.din, K = a.K, act = a.act;
  const int KB = (K + 15) / 16, K4 = pm_up(K, 4);
  real* gbuf = reinterpret_cast<real*>(pm_smem);       // dL/dout tile [32][PG]
  real* tbuf = gbuf + PM_RT * PG;                      // top hidden layer, then its pre-activation gradient
  real* lbuf = NL == 2 ? tbuf + PM_RT * PH : tbuf;     // first hidden layer (two layers)
  real* xbuf = tbuf + NL * PM_RT * PH;
  real* wbuf = xbuf + PM_RT * PX;
  const int t = threadIdx.x, l = t & 63, w = t >> 6, l16 = l & 15, q = l >> 4;
  const int64_t ntiles = (a.N + PM_RT - 1) / PM_RT;
  const real* W1 = a.param;
  const real* W2 = W1 + (int64_t)H * din + H;
  const real* W3 = NL == 2 ? W2 + (int64_t)H * H + H : W2;
  const real* htop = NL == 2 ? a.h2 : a.h1;

  acc_t acc3[4][NBH], acc1[NBH][DB];
  acc_t acc2[NL == 2 ? HB : 1][NBH];
  real sb1 = 0, sb2 = 0, sb3 = 0;
#pragma unroll
  for (int j = 0; j < NBH; ++j) {
#pragma unroll
    for (int k = 0; k < 4; ++k) acc3[k][j] = acc_t{0, 0, 0, 0};
#pragma unroll
    for (int d = 0; d < DB; ++d) acc1[j][d] = acc_t{0, 0, 0, 0};
#pragma unroll
    for (int o = 0; o < (NL == 2 ? HB : 1); ++o) acc2[o][j] = acc_t{0, 0, 0, 0};
  }
  v2 regs[NPW];
  // rows [j0, j0 + KC2) of W [rows][H] -> registers
  auto fetch = [&](const real* W, int rows, int j0) {
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int p = t + PM_BT * i, jj = p / (H / 2), u = 2 * (p - jj * (H / 2));
      v2 v = {real(0), real(0)};
      if (j0 + jj < rows) v = *reinterpret_cast<const v2*>(W + (int64_t)(j0 + jj) * H + u);
      regs[i] = v;
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int p = t + PM_BT * i, jj = p / (H / 2), u = 2 * (p - jj * (H / 2));
      *reinterpret_cast<v2*>(wbuf + jj * PW2 + u) = regs[i];
    }
  };
  // in-gradient of a hidden layer: img <- (A W) * act'(img), A = rows of `src`
  // (pitch PS, K dimension `kd` padded to kd4), W [kd][H]
  auto in_grad = [&](const real* src, int PS, const real* W, int kd, int kd4, real* img,
                     const real* nextW, int next_rows) {
    acc_t c[2][NBH];
#pragma unroll
    for (int j = 0; j < NBH; ++j) c[0][j] = c[1][j] = acc_t{0, 0, 0, 0};
    const int nch = (kd4 + KC2 - 1) / KC2;
#pragma unroll 1
    for (int ch = 0; ch < nch; ++ch) {
      __syncthreads();
      stash();
      __syncthreads();
      if (ch + 1 < nch) fetch(W, kd, (ch + 1) * KC2);
      else fetch(nextW, next_rows, 0);
      const int j0 = ch * KC2, ns = tmin(KC2, kd4 - j0) / 4;
#pragma unroll 2
      for (int ks = 0; ks < ns; ++ks) {
        const int k = j0 + ks * 4 + q;
        const real a0 = src[l16 * PS + k], a1 = src[(16 + l16) * PS + k];
#pragma unroll
        for (int j = 0; j < NBH; ++j) {
          const real b = wbuf[(ks * 4 + q) * PW2 + (w + 4 * j) * 16 + l16];
          c[0][j] = pmma(a0, b, c[0][j]);
          c[1][j] = pmma(a1, b, c[1][j]);
        }
      }
    }
    // (every wave passed >= 2 barriers since it last read `img` as an operand of
    // the preceding weight gradient; each lane rewrites the elements it reads)
#pragma unroll
    for (int j = 0; j < NBH; ++j)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          real* e = img + (rb * 16 + pm_drow<real>(q, i)) * PH + (w + 4 * j) * 16 + l16;
          *e = c[rb][j][i] * pm_act_d<real>(act, *e);
        }
    __syncthreads();
  };
  // column sums of an image over the 32 rows (thread = column)
  auto col_sum = [&](const real* img, int P, int ncol) -> real {
    real s = 0;
    if (t < ncol)
#pragma unroll 8
      for (int r = 0; r < PM_RT; ++r) s += img[r * P + t];
    return s;
  };

  fetch(W3, K, 0);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t r0 = tile * PM_RT;
    __syncthreads();                                   // the previous tile's readers
    for (int e = t; e < PM_RT * PG; e += PM_BT) {
      const int r = e / PG, c = e - r * PG;
      real v = 0;
      if (r0 + r < a.N && c < K) v = a.g[(r0 + r) * K + c];
      gbuf[e] = v;
    }
    for (int e = t; e < PM_RT * PX; e += PM_BT) {
      const int r = e / PX, c = e - r * PX;
      real v = 0;
      if (r0 + r < a.N && c < din) v = a.x[(r0 + r) * a.x_stride + c];
      xbuf[e] = v;
    }
    for (int p = t; p < PM_RT * (H / 2); p += PM_BT) {
      const int r = p / (H / 2), u = 2 * (p - r * (H / 2));
      v2 v = {real(0), real(0)}, v1 = v;
      if (r0 + r < a.N) {
        v = *reinterpret_cast<const v2*>(htop + (r0 + r) * H + u);
        if (NL == 2) v1 = *reinterpret_cast<const v2*>(a.h1 + (r0 + r) * H + u);
      }
      *reinterpret_cast<v2*>(tbuf + r * PH + u) = v;
      if (NL == 2) *reinterpret_cast<v2*>(lbuf + r * PH + u) = v1;
    }
    __syncthreads();
    // ---- dW3 += g^T top, db3
#pragma unroll 2
    for (int s = 0; s < 8; ++s) {
      const int ro = s + 8 * q;
      real bv[NBH];
#pragma unroll
      for (int j = 0; j < NBH; ++j) bv[j] = tbuf[ro * PH + (w + 4 * j) * 16 + l16];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        if (kb < KB) {
          const real av = gbuf[ro * PG + kb * 16 + l16];
#pragma unroll
          for (int j = 0; j < NBH; ++j) acc3[kb][j] = pmma(av, bv[j], acc3[kb][j]);
        }
      }
    }
    sb3 += col_sum(gbuf, PG, K);
    // ---- top <- (g W3) * act'(top)
    in_grad(gbuf, PG, W3, K, K4, tbuf, NL == 2 ? W2 : W3, NL == 2 ? H : K);
    if (NL == 2) {
      // ---- dW2 += dpre2^T h1, db2
#pragma unroll 2
      for (int s = 0; s < 8; ++s) {
        const int ro = s + 8 * q;
        real bv[NBH];
#pragma unroll
        for (int j = 0; j < NBH; ++j) bv[j] = lbuf[ro * PH + (w + 4 * j) * 16 + l16];
#pragma unroll
        for (int ob = 0; ob < HB; ++ob) {
          const real av = tbuf[ro * PH + ob * 16 + l16];
#pragma unroll
          for (int j = 0; j < NBH; ++j) acc2[ob][j] = pmma(av, bv[j], acc2[ob][j]);
        }
      }
      sb2 += col_sum(tbuf, PH, H);
      // ---- h1 <- (dpre2 W2) * act'(h1)
      in_grad(tbuf, PH, W2, H, H, lbuf, W3, K);
    }
    // ---- dW1 += dpre1^T x, db1
#pragma unroll 2
    for (int s = 0; s < 8; ++s) {
      const int ro = s + 8 * q;
      real bv[DB];
#pragma unroll
      for (int d = 0; d < DB; ++d) bv[d] = xbuf[ro * PX + d * 16 + l16];
#pragma unroll
      for (int j = 0; j < NBH; ++j) {
        const real av = lbuf[ro * PH + (w + 4 * j) * 16 + l16];
#pragma unroll
        for (int d = 0; d < DB; ++d) acc1[j][d] = pmma(av, bv[d], acc1[j][d]);
      }
    }
    sb1 += col_sum(lbuf, PH, H);
  }
  // ---- this workgroup's gradient slab, parameter order
  real* slab = a.partials + (int64_t)blockIdx.x * pm_num_params<H, NL>(din, K);
#pragma unroll
  for (int j = 0; j < NBH; ++j)
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int u = (w + 4 * j) * 16 + pm_drow<real>(q, i), c = d * 16 + l16;
        if (c < din) slab[(int64_t)u * din + c] = acc1[j][d][i];
      }
  if (t < H) slab[(int64_t)H * din + t] = sb1;
  int64_t off = (int64_t)H * din + H;
  if (NL == 2) {
#pragma unroll
    for (int ob = 0; ob < HB; ++ob)
#pragma unroll
      for (int j = 0; j < NBH; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          slab[off + (int64_t)(ob * 16 + pm_drow<real>(q, i)) * H + (w + 4 * j) * 16 + l16] =
              acc2[ob][j][i];
    if (t < H) slab[off + (int64_t)H * H + t] = sb2;
    off += (int64_t)H * H + H;
  }
#pragma unroll
  for (int kb = 0; kb < 4; ++kb)
#pragma unroll
    for (int j = 0; j < NBH; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = kb * 16 + pm_drow<real>(q, i);
        if (k < K) slab[off + (int64_t)k * H + (w + 4 * j) * 16 + l16] = acc3[kb][j][i];
      }
  if (t < K) slab[off + (int64_t)K * H + t] = sb3;
}

// grad[p] = sum over the slabs in a fixed order (8 loads in flight)
template <typename real>
__global__ __launch_bounds__(256) void pmlp_reduce_kernel(const real* __restrict__ part, int nb,
                                                          int64_t P, real* __restrict__ grad) {
  const int64_t p = blockIdx.x * 256ll + threadIdx.x;
  if (p >= P) return;
  real a8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int b = 0;
  for (; b + 8 <= nb; b += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a8[u] += part[(int64_t)(b + u) * P + p];
  }
  for (; b < nb; ++b) a8[0] += part[(int64_t)b * P + p];
  grad[p] = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
}

// ---- host side -------------------------------------------------------------------
inline int pm_cus() {
  static int n = 0;
  if (!n) {
    hipDeviceProp_t p;
    int dev = 0;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess)
            ? p.multiProcessorCount : 256;
  }
  return n;
}
// persistent workgroups of the backward launch: one per compute unit that is
// free (tce_set_cu_budget: beside the critic's grid), at most 128 slabs
inline int pm_bwd_grid(int64_t N) {
  const int64_t tiles = (N + PM_RT - 1) / PM_RT;
  const int budget = tce_cu_budget_value();
  const int cap = budget > 0 ? tmin(budget, 128) : 128;
  return (int)tmin<int64_t>(tiles, cap);
}
inline int pm_fwd_grid(int64_t N) {
  const int64_t tiles = (N + PM_RT - 1) / PM_RT;
  const int budget = tce_cu_budget_value();
  const int cap = 2 * (budget > 0 ? budget : pm_cus());
  return (int)tmin<int64_t>(tiles, cap);
}

inline bool pm_aligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline bool pm_shape_ok(int din, int H, int NL, int K, int elem) {
  if (din < 1 || din > 64 || K < 1 || K > 64) return false;
  if (elem == 4) return (H == 128 && (NL == 1 || NL == 2)) || (H == 256 && NL == 1);
  if (elem == 8) return (H == 128 && (NL == 1 || NL == 2)) || (H == 256 && NL == 1);
  return false;
}

template <typename real, int H, int NL>
int pm_forward_t(const PmArgs<real>& a, hipStream_t st) {
  const size_t lds = pm_fwd_lds<real, H>(a.din);
  auto kern = pmlp_fwd_kernel<real, H, NL>;
  tce_lds_limit(reinterpret_cast<const void*>(kern), lds);
  hipLaunchKernelGGL(kern, dim3(pm_fwd_grid(a.N)), dim3(PM_BT), lds, st, a);
  TCE_LAUNCH_CHECK();
  return 0;
}
template <typename real, int H, int NL, int DB>
int pm_backward_t(const PmArgs<real>& a, real* grad, hipStream_t st) {
  const size_t lds = pm_bwd_lds<real, H, NL, DB>();
  auto kern = pmlp_bwd_kernel<real, H, NL, DB>;
  tce_lds_limit(reinterpret_cast<const void*>(kern), lds);
  const int grid = pm_bwd_grid(a.N);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(PM_BT), lds, st, a);
  TCE_LAUNCH_CHECK();
  const int64_t P = pm_num_params<H, NL>(a.din, a.K);
  hipLaunchKernelGGL(pmlp_reduce_kernel<real>, dim3((unsigned)ceil_div(P, 256)), dim3(256), 0, st,
                     a.partials, grid, P, grad);
  TCE_LAUNCH_CHECK();
  return 0;
}

template <typename real>
int pm_forward(const real* x, int64_t x_stride, int64_t N, int din, int H, int NL, int K, int act,
               const real* param, real* h1, real* h2, real* out, hipStream_t st) {
  TCE_CHECK_ARG(x && param && out && N > 0, "pmlp_forward: null buffer / no rows");
  TCE_CHECK_ARG(pm_shape_ok(din, H, NL, K, (int)sizeof(real)) && act >= 0 && act <= 3,
                "pmlp_forward: shape not built (tce_pmlp_supported)");
  TCE_CHECK_ARG(x_stride >= din, "pmlp_forward: x_stride < din");
  TCE_CHECK_ARG(pm_aligned(param) && pm_aligned(h1) && pm_aligned(h2),
                "pmlp_forward: param / h1 / h2 must be 16-byte aligned");
  PmArgs<real> a{x, x_stride, N, din, K, act, param, h1, h2, out, nullptr, nullptr, {}};
  {
    const real* p = param;
    a.lay[0] = {p, p + (int64_t)H * din, h1, din, H};
    p += (int64_t)H * din + H;
    if (NL == 2) {
      a.lay[1] = {p, p + (int64_t)H * H, h2, H, H};
      p += (int64_t)H * H + H;
    }
    a.lay[NL] = {p, p + (int64_t)K * H, nullptr, H, K};
  }
  if (H == 128 && NL == 2) return pm_forward_t<real, 128, 2>(a, st);
  if (H == 128) return pm_forward_t<real, 128, 1>(a, st);
  return pm_forward_t<real, 256, 1>(a, st);
}
template <typename real>
int pm_backward(const real* x, int64_t x_stride, int64_t N, int din, int H, int NL, int K, int act,
                const real* param, const real* h1, const real* h2, const real* g, real* partials,
                real* grad, hipStream_t st) {
  TCE_CHECK_ARG(x && param && h1 && g && partials && grad && N > 0 && (NL == 1 || h2),
                "pmlp_backward: null buffer / no rows");
  TCE_CHECK_ARG(pm_shape_ok(din, H, NL, K, (int)sizeof(real)) && act >= 0 && act <= 3,
                "pmlp_backward: shape not built (tce_pmlp_supported)");
  TCE_CHECK_ARG(x_stride >= din, "pmlp_backward: x_stride < din");
  TCE_CHECK_ARG(pm_aligned(param) && pm_aligned(h1) && pm_aligned(h2),
                "pmlp_backward: param / h1 / h2 must be 16-byte aligned");
  PmArgs<real> a{x, x_stride, N, din, K, act, param, const_cast<real*>(h1),
                 const_cast<real*>(h2), nullptr, g, partials, {}};
  const bool small = din <= 32;
  if (H == 128 && NL == 2)
    return small ? pm_backward_t<real, 128, 2, 2>(a, grad, st)
                 : pm_backward_t<real, 128, 2, 4>(a, grad, st);
  if (H == 128)
    return small ? pm_backward_t<real, 128, 1, 2>(a, grad, st)
                 : pm_backward_t<real, 128, 1, 4>(a, grad, st);
  return small ? pm_backward_t<real, 256, 1, 2>(a, grad, st)
               : pm_backward_t<real, 256, 1, 4>(a, grad, st);
}

}  // namespace

extern "C" {

int tce_pmlp_supported(int din, int hidden, int num_hidden, int dout, int elem_size) {
  return pm_shape_ok(din, hidden, num_hidden, dout, elem_size) ? 1 : 0;
}
int64_t tce_pmlp_num_params(int din, int hidden, int num_hidden, int dout) {
  return (int64_t)hidden * din + hidden +
         (num_hidden == 2 ? (int64_t)hidden * hidden + hidden : 0) + (int64_t)dout * hidden + dout;
}
int tce_pmlp_max_slabs(void) { return 128; }

int tce_pmlp_forward_f32(const float* x, int64_t x_stride, int64_t N, int din, int hidden,
                         int num_hidden, int dout, int act, const float* param, float* h1,
                         float* h2, float* out, void* stream) {
  return pm_forward<float>(x, x_stride, N, din, hidden, num_hidden, dout, act, param, h1, h2, out,
                           (hipStream_t)stream);
}
int tce_pmlp_forward_f64(const double* x, int64_t x_stride, int64_t N, int din, int hidden,
                         int num_hidden, int dout, int act, const double* param, double* h1,
                         double* h2, double* out, void* stream) {
  return pm_forward<double>(x, x_stride, N, din, hidden, num_hidden, dout, act, param, h1, h2,
                            out, (hipStream_t)stream);
}
int tce_pmlp_backward_f32(const float* x, int64_t x_stride, int64_t N, int din, int hidden,
                          int num_hidden, int dout, int act, const float* param, const float* h1,
                          const float* h2, const float* grad_out, float* partials, float* grad,
                          void* stream) {
  return pm_backward<float>(x, x_stride, N, din, hidden, num_hidden, dout, act, param, h1, h2,
                            grad_out, partials, grad, (hipStream_t)stream);
}
int tce_pmlp_backward_f64(const double* x, int64_t x_stride, int64_t N, int din, int hidden,
                          int num_hidden, int dout, int act, const double* param,
                          const double* h1, const double* h2, const double* grad_out,
                          double* partials, double* grad, void* stream) {
  return pm_backward<double>(x, x_stride, N, din, hidden, num_hidden, dout, act, param, h1, h2,
                             grad_out, partials, grad, (hipStream_t)stream);
}

}  // extern "C"
