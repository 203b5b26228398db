// Small two-hidden-layer networks (D_in <= 64 -> H -> H -> D_out, H in {32, 64},
// fp32) of the black-box agent, one workgroup per 64 rows:
//
//   tce_smlp_forward_f32        values / means of a rollout
//                               (mprl/util/util_nn.py:225-246)
//   tce_smlp_critic_epochs_f32  E full-batch critic epochs, ONE launch each:
//                               forward + (clipped) value loss + backward +
//                               gradient reduction + grad-norm clip + Adam
//                               (mprl/rl/agent/black_box_agent.py:105-157,
//                               value_loss :391-419, grad_norm_clip
//                               mprl/util/util_numerical.py:244-275, Adam
//                               mprl/rl/agent/abstract_agent.py:62-82)
//   tce_bb_policy_epochs_f32    E policy epochs of the black-box agent with a
//                               shared (non-contextual) covariance, SIX launches
//                               each (black_box_agent.py:159-389): Cholesky head,
//                               covariance projection, the row kernel below,
//                               K x K KL parts, covariance projection backward,
//                               finish (Cholesky head backward + clip + Adam +
//                               record row).
//
// Row kernel.  Lane = row, the four waves of the workgroup split the hidden
// units (wave w owns units [w H/4, (w+1) H/4)): a layer is, per input unit, one
// broadcast read of the weights of the wave's output units from the LDS image
// and H/4 FMAs; activations pass between the layers through row-major LDS
// tiles (pitch 4 x odd: a lane's 16-byte pieces are conflict free).  These
// nets are far too small for the matrix cores to matter (56 MFLOP per epoch at
// 4096 rows; the exact-fp32 MFMA rate equals the packed VALU rate): what counts
// is the length of the dependent chain, so the whole epoch is one launch.
// Between forward and backward wave 0 runs the "head" per row: the value loss,
// or -- policy -- mean projection, log-prob of the sampled parameters under the
// projected Gaussian, surrogate gradient, trust-region gradient and the way
// back through the mean projection (triangular solves against the K x K
// factors held in LDS, vectors in [k][lane] layout).  Weight gradients are
// sums of outer products over the 64 rows: thread = 4 x 4 block of a weight
// matrix, operands read as 16-byte pieces of the row-major tiles.  Every
// workgroup writes its gradient slab; a second, wide launch adds the slabs in a
// fixed order (deterministic) and -- critic without clipping -- applies the Adam
// step in the same pass.  (One workgroup adding 64 slabs after the last ticket
// took 250 us: 600 dependent-latency loads per thread; 38 workgroups x 1024
// threads need one round trip.)
//
// Diagonal covariances (std_only: the reference's BBRL configuration) make
// every K x K step a K-vector step: one small kernel does Cholesky head,
// covariance projection, KL parts and the trust region gradient, and the
// finish kernel does the projection's backward -- four launches per epoch
// instead of seven.
#include "mlp_shared.h"
#include "xchg.h"
#include "smallmat.h"
#include "../../include/tce_hip.h"

namespace {

constexpr int SR = 64;                  // rows per workgroup
constexpr int SBT = 256;                // threads per workgroup
constexpr int SNW = SBT / 64;
constexpr float S_HALF_LOG_2PI = 0.9189385332046727f;
constexpr size_t S_LDS_MAX = 160 * 1024;
constexpr int S_MAX_GRID = 1024;

__host__ __device__ inline int s_up4(int n) { return (n + 3) & ~3; }
// pitch (floats) of a row-major [64][w] tile: 4 x odd, so that the 16-byte
// pieces of 16 consecutive rows fall into 16 different bank groups
__host__ __device__ inline int s_pitch(int w) { return 4 * (((w + 3) / 4) | 1); }
// pitch of the K x K factor images in LDS: K padded to the length the register
// head is unrolled over (rows / columns past K are zero, so no bounds tests)
__host__ __device__ inline int s_kpad(int K) {
  return K <= 8 ? 8 : K <= 16 ? 16 : K <= 24 ? 24 : K <= 32 ? 32 : s_up4(K);
}
__host__ __device__ inline int s_nparams(int din, int H, int dout) {
  return H * din + H + H * H + H + dout * H + dout;
}

enum { HEAD_NONE = 0, HEAD_VALUE = 1, HEAD_BB_POLICY = 2 };

// ACT_RT: the activation is a kernel argument (the policy kernels, which come
// in one variant per padded K instead of one per activation)
constexpr int ACT_RT = 4;
template <int ACT>
__device__ inline float s_actf(float y, int rt) {
  if (ACT != ACT_RT) return act_f < ACT == ACT_RT ? 0 : ACT > (y);
  switch (rt) {
    case ACT_TANH: return act_f<ACT_TANH>(y);
    case ACT_RELU: return act_f<ACT_RELU>(y);
    case ACT_LEAKY: return act_f<ACT_LEAKY>(y);
    default: return act_f<ACT_SOFTPLUS>(y);
  }
}
template <int ACT>
__device__ inline float s_actd(float h, int rt) {
  if (ACT != ACT_RT) return act_d < ACT == ACT_RT ? 0 : ACT > (h);
  switch (rt) {
    case ACT_TANH: return act_d<ACT_TANH>(h);
    case ACT_RELU: return act_d<ACT_RELU>(h);
    case ACT_LEAKY: return act_d<ACT_LEAKY>(h);
    default: return act_d<ACT_SOFTPLUS>(h);
  }
}

// diagnostic build (scripts/smlp_stamps.py): cycles per phase of the row kernel;
// SMLP_STAMP_HEAD: the sections of the register policy head instead
#ifdef SMLP_STAMP
#define SMLP_T0() long long t0_ = __builtin_readcyclecounter(); long long stamp_[5] = {0, 0, 0, 0, 0};
#define SMLP_T(k) { const long long tn_ = __builtin_readcyclecounter(); stamp_[k] += tn_ - t0_; t0_ = tn_; }
#else
#define SMLP_T0()
#define SMLP_T(k)
#endif
#ifdef SMLP_STAMP_HEAD
#define SMLP_HARGS_DECL , long long* stamp_, long long& t0_
#define SMLP_HARGS , stamp_, t0_
#define SMLP_TH(k) SMLP_T(k)
#define SMLP_TK(k)
#else
#define SMLP_HARGS_DECL
#define SMLP_HARGS
#define SMLP_TH(k)
#define SMLP_TK(k) SMLP_T(k)
#endif

// LDS map (offsets in floats, all multiples of 4)
struct SLds {
  int w1t, w2, w2t, w3, b1, b2, b3;      // weight images
  int xs, h1s, h2s, d1s, d2s, g3s;       // row-major tiles [64][pitch]
  int ys, gus;                           // policy: y and g u (row-major [64][gp])
  int vec;                               // policy: 6 vectors [doutp][64]; value head: 2 x [64]
  int lo, lp, rdo, rdp;                  // policy: L_old, L_proj [kp][kp], 1 / diagonals
  int lio, liot, lip, lipt;              // policy: L_old^-1, its transpose, L_proj^-1, transpose
  int red;                               // scratch
  int total;
  int xp, hp, gp, dinp, doutp;
};
__host__ __device__ inline SLds s_lds(int din, int H, int dout, int head) {
  SLds L;
  L.dinp = s_up4(din);
  L.doutp = s_up4(dout);
  L.xp = s_pitch(din);
  L.hp = s_pitch(H);
  L.gp = s_pitch(dout);
  int o = 0;
  L.w1t = o; o += L.dinp * H;
  L.w2 = o; o += H * H;
  L.w2t = o; o += H * H;
  L.w3 = o; o += L.doutp * H;
  L.b1 = o; o += H;
  L.b2 = o; o += H;
  L.b3 = o; o += L.doutp;
  L.xs = o; o += SR * L.xp;
  L.h1s = o; o += SR * L.hp;
  L.h2s = o; o += SR * L.hp;
  const bool bwd = head != HEAD_NONE;
  L.d1s = o; o += bwd ? SR * L.hp : 0;
  L.d2s = o; o += bwd ? SR * L.hp : 0;
  L.g3s = o; o += bwd ? SR * L.gp : 0;
  const bool pol = head == HEAD_BB_POLICY;
  L.ys = o; o += pol ? SR * L.gp : 0;
  L.gus = o; o += pol ? SR * L.gp : 0;
  L.vec = o; o += pol ? 6 * L.doutp * SR : 2 * SR;
  const int kp = s_kpad(dout);
  const int kk = kp * kp;
  L.lo = o; o += pol ? kk : 0;
  L.lp = o; o += pol ? kk : 0;
  L.rdo = o; o += pol ? kp : 0;           // (kp entries: the diagonal head reads its padded length)
  L.rdp = o; o += pol ? kp : 0;
  const bool inv = pol && kp <= 32;       // the register head (K <= 32) works with the inverses
  L.lio = o; o += inv ? kk : 0;
  L.liot = o; o += inv ? kk : 0;
  L.lip = o; o += inv ? kk : 0;
  L.lipt = o; o += inv ? kk : 0;
  L.red = o; o += 16;
  L.total = o;
  return L;
}

struct SNet {
  const float* x;            // rows: x + r * x_stride
  int64_t x_stride;
  int64_t N;
  int din, dout;
  const float* param;        // flat: W1 [H][din] | b1 | W2 [H][H] | b2 | W3 [dout][H] | b3
  int act;                   // activation (read by the ACT_RT variants)
};

struct SValueHead {
  const float* ret;
  const float* old_v;        // nullable unless clip > 0
  float clip;
};

struct SPolicyHead {
  const float *actions, *logp_old, *adv, *mean_old;   // [N,K], [N], [N], [N,K]
  const float *L_old, *L_proj;                        // [K,K] each (shared)
  const float *Li_old, *Li_proj;                      // their inverses [K,K]
  float eps_mean, tr_coeff, ent_coef;
  float *mean_out, *pmean_out;                        // nullable [N,K]
};

struct SReduce {
  float* slabs;              // [grid][PS]: network gradient | (policy) sum g u y^T [K][K]
  double* dpart;             // [grid][8]: value {loss sum}; policy {sum ratio adv, sum ratio, m1, m2, m3}
  int PS;                    // slab pitch
  int P;                     // network parameters
};

// arguments of the slab reduction (smlp_reduce_kernel)
struct SFinish {
  const float* slabs;
  const double* dpart;
  int nparts, PS, P, KK;     // KK: extra columns behind the network's (policy: K * K)
  int64_t N;
  float* grad;               // [P]
  float* extra;              // [KK] (policy: raw sum g u y^T)
  double* dsum;              // [8] sums of dpart over the workgroups
  float* stats;              // {mean loss (value head), |grad|^2 (+= : zeroed by the caller)}
  // Adam fused into the reduction (value head without clipping); param == nullptr: gradient only
  float *param, *m, *v, *state;
  float lr, b1, b2, eps, wd, step, gscale;
};

// ---------------------------------------------------------------------------
template <int H>
__device__ inline void s_load_weights(const SLds& L, float* S, const SNet& n) {
  const float* W1 = n.param;
  const float* B1 = W1 + H * n.din;
  const float* W2 = B1 + H;
  const float* B2 = W2 + H * H;
  const float* W3 = B2 + H;
  const float* B3 = W3 + n.dout * H;
  const int tid = threadIdx.x;
  for (int e = tid; e < L.dinp * H; e += SBT) {
    const int i = e / H, j = e - i * H;
    S[L.w1t + e] = i < n.din ? W1[j * n.din + i] : 0.f;
  }
  for (int e = tid; e < H * H; e += SBT) {
    const float w = W2[e];
    const int j = e / H, i = e - j * H;
    S[L.w2 + e] = w;
    S[L.w2t + i * H + j] = w;
  }
  for (int e = tid; e < L.doutp * H; e += SBT) S[L.w3 + e] = e < n.dout * H ? W3[e] : 0.f;
  for (int e = tid; e < H; e += SBT) {
    S[L.b1 + e] = B1[e];
    S[L.b2 + e] = B2[e];
  }
  for (int e = tid; e < L.doutp; e += SBT) S[L.b3 + e] = e < n.dout ? B3[e] : 0.f;
}

__device__ inline void s_load_x(const SLds& L, float* S, const SNet& n, int64_t r0) {
  for (int e = threadIdx.x; e < SR * L.dinp; e += SBT) {
    const int r = e / L.dinp, c = e - r * L.dinp;
    const int64_t row = r0 + r;
    float v = 0.f;
    if (c < n.din && row < n.N) v = n.x[row * n.x_stride + c];
    S[L.xs + r * L.xp + c] = v;
  }
}

// The same images by LDS-DMA (global_load_lds_dword: one wave instruction moves
// 64 floats from per-lane global addresses to 64 consecutive LDS floats): no
// data registers, no store pass, and every load of the weights AND of the
// first tile is in flight before anything is waited for -- ONE round trip to
// memory, where the loops above are a dependent round trip per image (23 000
// of the policy row kernel's 81 000 cycles at 4096 rows).  (Staged through
// register arrays instead, the compiler copied the loaded values to AGPRs as
// they arrived -- a wait per batch -- and the predicates were 4 000
// instructions.)  Chunk t of an image = its floats [64 t, 64 t + 64), lane =
// float; the waves take chunks in turn; the source address is clamped instead
// of predicated (pad columns / rows past N hold copies of valid elements:
// the zero pad rows of W1^T, written behind the wait, and the heads' row masks
// make them harmless).  e / d for a run-time d and e < 2^15 / d is exact as
// (e * ceil(2^20 / d)) >> 20.
typedef const __attribute__((address_space(1))) void* s_gvp;
typedef __attribute__((address_space(3))) void* s_lvp;
__device__ __forceinline__ void s_dma64(const float* src, float* lds_chunk) {
  __builtin_amdgcn_global_load_lds((s_gvp)src, (s_lvp)lds_chunk, 4, 0, 0);
}
__device__ __forceinline__ unsigned s_magic(int d) { return ((1u << 20) + (unsigned)d - 1u) / (unsigned)d; }

template <int H>
__device__ __forceinline__ void s_dma_weights(const SLds& L, float* S, const SNet& n, int wave,
                                              unsigned lane) {
  const float* __restrict__ W1 = n.param;
  const float* __restrict__ B1 = W1 + H * n.din;
  const float* __restrict__ W2 = B1 + H;
  const float* __restrict__ B2 = W2 + H * H;
  const float* __restrict__ W3 = B2 + H;
  const float* __restrict__ B3 = W3 + n.dout * H;
  const unsigned din = (unsigned)n.din;
  // W1^T image [dinp][H]: float a = i H + j  <-  W1[j][min(i, din - 1)]
  for (int t = wave; t < L.dinp * H / 64; t += SNW) {
    const unsigned a = 64u * t + lane, i = a / H, j = a % H;
    s_dma64(W1 + j * din + (i < din ? i : din - 1u), S + L.w1t + 64 * t);
  }
  // W2 as is, and transposed: w2t[i H + j] = W2[j][i]
  for (int t = wave; t < H * H / 64; t += SNW) {
    const unsigned a = 64u * t + lane;
    s_dma64(W2 + a, S + L.w2 + 64 * t);
    s_dma64(W2 + (a % H) * H + a / H, S + L.w2t + 64 * t);
  }
  const unsigned n3 = (unsigned)(n.dout * H) - 1u;
  for (int t = wave; t < L.doutp * H / 64; t += SNW) {
    const unsigned a = 64u * t + lane;
    s_dma64(W3 + (a < n3 ? a : n3), S + L.w3 + 64 * t);
  }
  // (H <= 64, doutp <= 64: one masked chunk each)
  if (wave == 0 && lane < H) s_dma64(B1 + lane, S + L.b1);
  if (wave == 1 && lane < H) s_dma64(B2 + lane, S + L.b2);
  if (wave == 2 && lane < (unsigned)L.doutp)
    s_dma64(B3 + (lane < (unsigned)n.dout ? lane : (unsigned)n.dout - 1u), S + L.b3);
}
// behind the wait for the DMA: the pad rows of W1^T (inputs din .. dinp - 1) are zero
template <int H>
__device__ __forceinline__ void s_zero_w1_pad(const SLds& L, float* S, const SNet& n) {
  for (int e = n.din * H + threadIdx.x; e < L.dinp * H; e += SBT) S[L.w1t + e] = 0.f;
}

// a tile: x rows into [64][xp] (pad columns: copies of the last input), and --
// policy -- the old means and the sampled parameters into [k][64]
template <bool POL, bool ROWMAJOR>
__device__ __forceinline__ void s_dma_tile(const SLds& L, float* S, const SNet& n,
                                           const SPolicyHead& ph, int64_t r0, int wave,
                                           unsigned lane) {
  const unsigned last = (unsigned)(n.N - r0 < SR ? n.N - r0 : SR) - 1u;   // last valid row of the tile
  const float* __restrict__ xb = n.x + r0 * n.x_stride;
  const unsigned stride = (unsigned)n.x_stride, xp = (unsigned)L.xp, cl = (unsigned)n.din - 1u;
  const unsigned mx = s_magic(L.xp);
  for (int t = wave; t < L.xp; t += SNW) {                   // 64 xp floats = xp chunks
    const unsigned a = 64u * t + lane, r = (a * mx) >> 20, c = a - r * xp;
    s_dma64(xb + (r < last ? r : last) * stride + (c < cl ? c : cl), S + L.xs + 64 * t);
  }
  if (POL) {
    const unsigned K = (unsigned)n.dout;
    const float* __restrict__ mob = ph.mean_old + r0 * K;
    const float* __restrict__ acb = ph.actions + r0 * K;
    const int VS = L.doutp * SR;
    if (ROWMAJOR) {
      // the diagonal head reads them as rows: the tile's [64][K] blocks as they are
      // (coalesced), and logp_old / adv of the rows into slot 4
      const unsigned top = (last + 1u) * K - 1u;
      for (int t = wave; t < n.dout; t += SNW) {             // 64 K floats = K chunks
        const unsigned a = 64u * t + lane, o = a < top ? a : top;
        s_dma64(mob + o, S + L.vec + VS + 64 * t);
        s_dma64(acb + o, S + L.vec + 2 * VS + 64 * t);
      }
      const unsigned rr = lane < last ? lane : last;
      if (wave == 3) s_dma64(ph.logp_old + r0 + rr, S + L.vec + 4 * VS);
      if (wave == 2) s_dma64(ph.adv + r0 + rr, S + L.vec + 4 * VS + SR);
    } else {
      const unsigned ro = (lane < last ? lane : last) * K;
      for (int k = wave; k < n.dout; k += SNW) {
        s_dma64(mob + ro + k, S + L.vec + VS + k * SR);
        s_dma64(acb + ro + k, S + L.vec + 2 * VS + k * SR);
      }
    }
  }
}
// all LDS-DMA of this wave has landed (the barrier that follows publishes it)
__device__ __forceinline__ void s_dma_wait() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// forward of the 64 rows of the tile: h1 / h2 (this wave's H/4 units of its
// lane's row, after the activation) stay in registers for the backward pass,
// the output goes to out[o * 64 + lane] (LDS, [dout][64])
template <int H, int ACT>
__device__ inline void s_forward(const SLds& L, float* S, int dout, int lane, int wave,
                                 float (&h1)[H / SNW], float (&h2)[H / SNW], float* out,
                                 int rt_act) {
  constexpr int US = H / SNW;
  const int u0 = wave * US;
  float acc[US];
#pragma unroll
  for (int u = 0; u < US; ++u) acc[u] = S[L.b1 + u0 + u];
  for (int i0 = 0; i0 < L.dinp; i0 += 4) {
    const f32x4 xv = *reinterpret_cast<const f32x4*>(S + L.xs + lane * L.xp + i0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float* wr = S + L.w1t + (i0 + t) * H + u0;
#pragma unroll
      for (int u = 0; u < US; u += 4) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + u);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[u + q] += wv[q] * xv[t];
      }
    }
  }
#pragma unroll
  for (int u = 0; u < US; ++u) h1[u] = s_actf<ACT>(acc[u], rt_act);
#pragma unroll
  for (int u = 0; u < US; u += 4)
    *reinterpret_cast<f32x4*>(S + L.h1s + lane * L.hp + u0 + u) =
        f32x4{h1[u], h1[u + 1], h1[u + 2], h1[u + 3]};
  __syncthreads();
  float hin[H];
#pragma unroll
  for (int q = 0; q < H; q += 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(S + L.h1s + lane * L.hp + q);
    hin[q] = t[0]; hin[q + 1] = t[1]; hin[q + 2] = t[2]; hin[q + 3] = t[3];
  }
#pragma unroll
  for (int u = 0; u < US; ++u) acc[u] = S[L.b2 + u0 + u];
#pragma unroll
  for (int i = 0; i < H; ++i) {
    const float* wr = S + L.w2t + i * H + u0;
#pragma unroll
    for (int u = 0; u < US; u += 4) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + u);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[u + q] += wv[q] * hin[i];
    }
  }
#pragma unroll
  for (int u = 0; u < US; ++u) h2[u] = s_actf<ACT>(acc[u], rt_act);
#pragma unroll
  for (int u = 0; u < US; u += 4)
    *reinterpret_cast<f32x4*>(S + L.h2s + lane * L.hp + u0 + u) =
        f32x4{h2[u], h2[u + 1], h2[u + 2], h2[u + 3]};
  __syncthreads();
#pragma unroll
  for (int q = 0; q < H; q += 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(S + L.h2s + lane * L.hp + q);
    hin[q] = t[0]; hin[q + 1] = t[1]; hin[q + 2] = t[2]; hin[q + 3] = t[3];
  }
  for (int o = wave; o < dout; o += SNW) {
    const float* wr = S + L.w3 + o * H;
    float p[4] = {S[L.b3 + o], 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < H; j += 4) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + j);
#pragma unroll
      for (int q = 0; q < 4; ++q) p[q] += wv[q] * hin[j + q];
    }
    out[o * SR + lane] = (p[0] + p[1]) + (p[2] + p[3]);
  }
  __syncthreads();
}

// backward of the tile from g[o * 64 + lane] = dLoss / d out: the dY2 / dY1 tiles
template <int H, int ACT>
__device__ inline void s_backward(const SLds& L, float* S, int dout, int lane, int wave,
                                  const float (&h1)[H / SNW], const float (&h2)[H / SNW],
                                  const float* g, int rt_act) {
  constexpr int US = H / SNW;
  const int u0 = wave * US;
  float acc[US];
#pragma unroll
  for (int u = 0; u < US; ++u) acc[u] = 0.f;
  for (int o = 0; o < dout; ++o) {
    const float gv = g[o * SR + lane];
    const float* wr = S + L.w3 + o * H + u0;
#pragma unroll
    for (int u = 0; u < US; u += 4) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + u);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[u + q] += wv[q] * gv;
    }
  }
#pragma unroll
  for (int u = 0; u < US; u += 4) {
    f32x4 t;
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = acc[u + q] * s_actd<ACT>(h2[u + q], rt_act);
    *reinterpret_cast<f32x4*>(S + L.d2s + lane * L.hp + u0 + u) = t;
  }
  __syncthreads();
  float din_[H];
#pragma unroll
  for (int q = 0; q < H; q += 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(S + L.d2s + lane * L.hp + q);
    din_[q] = t[0]; din_[q + 1] = t[1]; din_[q + 2] = t[2]; din_[q + 3] = t[3];
  }
#pragma unroll
  for (int u = 0; u < US; ++u) acc[u] = 0.f;
#pragma unroll
  for (int j = 0; j < H; ++j) {
    const float* wr = S + L.w2 + j * H + u0;
#pragma unroll
    for (int u = 0; u < US; u += 4) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + u);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[u + q] += wv[q] * din_[j];
    }
  }
#pragma unroll
  for (int u = 0; u < US; u += 4) {
    f32x4 t;
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = acc[u + q] * s_actd<ACT>(h1[u + q], rt_act);
    *reinterpret_cast<f32x4*>(S + L.d1s + lane * L.hp + u0 + u) = t;
  }
  __syncthreads();
}

// out[(4 bi + u) * ldo + 4 bj + v] (+)= sum over the 64 rows of A[r][4 bi + u] B[r][4 bj + v]
// for the 4 x 4 block `blk` of an M x N matrix (A: [64][pa], B: [64][pb])
__device__ inline void s_outer_block(const float* A, int pa, const float* B, int pb, int M, int N,
                                     int blk, float* out, bool accumulate) {
  const int nbj = (N + 3) / 4;
  const int bi = blk / nbj, bj = blk - bi * nbj;
  float acc[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = 0.f;
  const float* a = A + 4 * bi;
  const float* b = B + 4 * bj;
#pragma unroll 16
  for (int r = 0; r < SR; ++r) {
    const f32x4 av = *reinterpret_cast<const f32x4*>(a + r * pa);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(b + r * pb);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[u][v] += av[u] * bv[v];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int i = 4 * bi + u, j = 4 * bj + v;
      if (i < M && j < N) {
        float* dst = out + i * N + j;
        *dst = accumulate ? *dst + acc[u][v] : acc[u][v];
      }
    }
}

// The same sums on the matrix cores (exact fp32 MFMA 16x16x4): unit = (weight
// matrix, block of 16 of its rows); per four tile rows one A fragment
// (A[r][m0 + i], a ds_read_b32 per lane), up to four B fragments and as many
// MFMAs, plus one MFMA against a fragment of ones -- the bias gradient (column
// sums of A) for free.  Fragments past a matrix's columns read whatever
// follows in the LDS (initialised tiles: finite) into outputs that are never
// stored.  The waves take the units in turn.  As 4 x 4 register blocks per
// thread (above) the pass was 1024 FMAs + 128 LDS reads per thread: 12 000 -
// 14 000 of the row kernel's cycles; this is ~3 000.
typedef float s_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void s_grad_unit(const float* A, int pa, const float* B, int pb, int M,
                                            int N, int m0, float* gW, float* gB, bool accumulate,
                                            int lane) {
  const int li = lane & 15, lk = lane >> 4;
  const int NT = (N + 15) >> 4;                               // <= 4
  s_f32x4 acc[4], accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (s_f32x4){0.f, 0.f, 0.f, 0.f};
  const float* a = A + lk * pa + m0 + li;
  const float* b = B + lk * pb + li;
#pragma unroll 4
  for (int r = 0; r < SR; r += 4) {
    const float av = a[r * pa];
    float bv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) bv[t] = t < NT ? b[r * pb + 16 * t] : 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (t < NT) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[t], acc[t], 0, 0, 0);
    if (gB) accb = __builtin_amdgcn_mfma_f32_16x16x4f32(av, 1.f, accb, 0, 0, 0);
  }
  // D: register i of lane l = D[4 (l / 16) + i][l % 16]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + 4 * lk + i;
    if (m < M) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int col = 16 * t + li;
        if (t < NT && col < N) {
          float* dst = gW + m * N + col;
          *dst = accumulate ? *dst + acc[t][i] : acc[t][i];
        }
      }
      if (gB && li == 0) gB[m] = accumulate ? gB[m] + accb[i] : accb[i];
    }
  }
}
template <int H>
__device__ inline void s_param_grads_mfma(const SLds& L, const float* S, int din, int dout,
                                          float* slab, bool accumulate, float* gpl, int wave,
                                          int lane) {
  float* gW1 = slab;
  float* gB1 = gW1 + H * din;
  float* gW2 = gB1 + H;
  float* gB2 = gW2 + H * H;
  float* gW3 = gB2 + H;
  float* gB3 = gW3 + dout * H;
  constexpr int MH = H / 16;                                  // row blocks of W1 / W2
  const int M3 = (dout + 15) >> 4;                            // of W3 / the K x K sum
  // units in an order that spreads the long ones (W1: up to 4 + 1 MFMAs per step)
  const int nu = 2 * MH + M3 + (gpl ? M3 : 0);
  for (int u = wave; u < nu; u += SNW) {
    if (u < MH)
      s_grad_unit(S + L.d1s, L.hp, S + L.xs, L.xp, H, din, 16 * u, gW1, gB1, accumulate, lane);
    else if (u < MH + M3 && gpl)
      s_grad_unit(S + L.gus, L.gp, S + L.ys, L.gp, dout, dout, 16 * (u - MH), gpl, nullptr,
                  accumulate, lane);
    else if (u < 2 * MH + (gpl ? M3 : 0))
      s_grad_unit(S + L.d2s, L.hp, S + L.h1s, L.hp, H, H, 16 * (u - MH - (gpl ? M3 : 0)), gW2,
                  gB2, accumulate, lane);
    else
      s_grad_unit(S + L.g3s, L.gp, S + L.h2s, L.hp, dout, H, 16 * (u - 2 * MH - (gpl ? M3 : 0)),
                  gW3, gB3, accumulate, lane);
  }
}

// the weight / bias gradients of the tile into `slab` (accumulate: a later
// tile of the same workgroup adds to it)
template <int H>
__device__ inline void s_param_grads(const SLds& L, const float* S, int din, int dout, float* slab,
                                     bool accumulate, float* gpl) {
  const int nb1 = (H / 4) * (L.dinp / 4), nb2 = (H / 4) * (H / 4), nb3 = (L.doutp / 4) * (H / 4);
  // policy head: d logp / d L_proj summed over the rows, sum g u y^T (gpl; the
  // lower triangle is used), as further blocks of the same pass
  const int nb4 = gpl ? (L.doutp / 4) * (L.doutp / 4) : 0;
  float* gW1 = slab;
  float* gB1 = gW1 + H * din;
  float* gW2 = gB1 + H;
  float* gB2 = gW2 + H * H;
  float* gW3 = gB2 + H;
  float* gB3 = gW3 + dout * H;
  for (int t = threadIdx.x; t < nb1 + nb2 + nb3 + nb4; t += SBT) {
    if (t < nb1)
      s_outer_block(S + L.d1s, L.hp, S + L.xs, L.xp, H, din, t, gW1, accumulate);
    else if (t < nb1 + nb2)
      s_outer_block(S + L.d2s, L.hp, S + L.h1s, L.hp, H, H, t - nb1, gW2, accumulate);
    else if (t < nb1 + nb2 + nb3)
      s_outer_block(S + L.g3s, L.gp, S + L.h2s, L.hp, dout, H, t - nb1 - nb2, gW3, accumulate);
    else
      s_outer_block(S + L.gus, L.gp, S + L.ys, L.gp, dout, dout, t - nb1 - nb2 - nb3, gpl,
                    accumulate);
  }
  for (int t = threadIdx.x; t < 2 * H + dout; t += SBT) {
    const float* src;
    int p;
    float* dst;
    if (t < H) { src = S + L.d1s + t; p = L.hp; dst = gB1 + t; }
    else if (t < 2 * H) { src = S + L.d2s + (t - H); p = L.hp; dst = gB2 + (t - H); }
    else { src = S + L.g3s + (t - 2 * H); p = L.gp; dst = gB3 + (t - 2 * H); }
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < SR; ++r) s += src[r * p];
    *dst = accumulate ? *dst + s : s;
  }
}

// ---------------------------------------------------------------------------
// value head (wave 0): loss and dLoss / dv of the tile's rows
// (black_box_agent.py:391-419 / temporal_correlated_agent.py:688-716)
__device__ inline double s_value_head(const SLds& L, float* S, const SValueHead& h, int64_t r0,
                                      int64_t N, int lane, const float* out, float* g) {
  const int64_t row = r0 + lane;
  const bool rok = row < N;
  const int64_t rc = rok ? row : N - 1;
  const float v = out[lane], rt = h.ret[rc];
  const float e = v - rt;
  float l = e * e, d = 2.f * e;
  if (h.clip > 0.f) {
    const float ov = h.old_v[rc];
    const float dlt = v - ov;
    const float cl = fminf(fmaxf(dlt, -h.clip), h.clip);
    const float e2 = ov + cl - rt;
    if (e2 * e2 > l) { l = e2 * e2; d = (dlt > -h.clip && dlt < h.clip) ? 2.f * e2 : 0.f; }
  }
  if (!rok) { l = 0.f; d = 0.f; }
  const float dv = d / (float)N;
  g[lane] = dv;
  S[L.g3s + lane * L.gp] = dv;
  return wave_sum_f64((double)l);
}

// policy head of the black-box agent (wave 0), per row: mean projection
// (third-party KL projection layer, oracle/kl_oracle.py mean_projection),
// log N(action; proj_mean, L_proj L_proj^T) (black_box_policy.py:95-128),
// surrogate gradient (black_box_agent.py:421-443), the mean part of the trust
// region loss and its gradient, and back through the mean projection.
// vec: MU | MO | AC | Z | Y | W, each [doutp][64].  Returns this wave's sums in
// `sums`: {sum ratio adv, sum ratio, m1, m2, m3}.
__device__ inline void s_policy_head(const SLds& L, float* S, const SPolicyHead& h, int64_t r0,
                                     int64_t N, int K, int lane, float logdet_p,
                                     double (&sums)[5]) {
  const int VS = L.doutp * SR;
  float* MU = S + L.vec;
  float* MO = MU + VS;
  float* AC = MO + VS;
  float* Z = AC + VS;
  float* Y = Z + VS;
  float* W = Y + VS;
  const float* Lo = S + L.lo;
  const float* Lp = S + L.lp;
  const float* rdo = S + L.rdo;
  const float* rdp = S + L.rdp;
  const int64_t row = r0 + lane;
  const bool rok = row < N;
  const int64_t rc = rok ? row : N - 1;
  const int LP = s_kpad(K);
  // z = L_old^-1 (mu - mu_old)
  float quad = 0.f;
  for (int r = 0; r < K; ++r) {
    float v = MU[r * SR + lane] - MO[r * SR + lane];
    for (int k = 0; k < r; ++k) v -= Lo[r * LP + k] * Z[k * SR + lane];
    v *= rdo[r];
    Z[r * SR + lane] = v;
    quad += v * v;
  }
  const float m = 0.5f * quad;
  const bool active = m > h.eps_mean;
  const float sc = active ? sqrtf(m / h.eps_mean) : 1.f;
  const float om = sc - 1.f;
  const float den = 1.f + om + 1e-16f;
  // y = L_proj^-1 (a - proj_mean), proj_mean = (mu + om mu_old) / (1 + om)
  float quady = 0.f;
  for (int r = 0; r < K; ++r) {
    const float mu = MU[r * SR + lane];
    const float pm = active ? (mu + om * MO[r * SR + lane]) / den : mu;
    if (h.pmean_out && rok) h.pmean_out[row * K + r] = pm;
    if (h.mean_out && rok) h.mean_out[row * K + r] = mu;
    float v = AC[r * SR + lane] - pm;
    for (int k = 0; k < r; ++k) v -= Lp[r * LP + k] * Y[k * SR + lane];
    v *= rdp[r];
    Y[r * SR + lane] = v;
    quady += v * v;
    // w (below) starts from mu - proj_mean
    W[r * SR + lane] = mu - pm;
  }
  const float logp = -0.5f * quady - logdet_p - S_HALF_LOG_2PI * (float)K;
  const float ratio = expf(logp - h.logp_old[rc]);
  const float ra = ratio * h.adv[rc];
  const float g = rok ? -ra / (float)N : 0.f;          // d surrogate / d logp
  for (int r = 0; r < K; ++r) S[L.ys + lane * L.gp + r] = Y[r * SR + lane];
  // u = L_proj^-T y (in place), d logp / d proj_mean = u
  for (int r = K - 1; r >= 0; --r) {
    float v = Y[r * SR + lane];
    for (int k = r + 1; k < K; ++k) v -= Lp[k * LP + r] * Y[k * SR + lane];
    Y[r * SR + lane] = v * rdp[r];
  }
  for (int r = 0; r < K; ++r) S[L.gus + lane * L.gp + r] = g * Y[r * SR + lane];
  // w = L_proj^-1 (mu - proj_mean), q = L_proj^-T w (in place)
  float maha2 = 0.f;
  for (int r = 0; r < K; ++r) {
    float v = W[r * SR + lane];
    for (int k = 0; k < r; ++k) v -= Lp[r * LP + k] * W[k * SR + lane];
    v *= rdp[r];
    W[r * SR + lane] = v;
    maha2 += v * v;
  }
  for (int r = K - 1; r >= 0; --r) {
    float v = W[r * SR + lane];
    for (int k = r + 1; k < K; ++k) v -= Lp[k * LP + r] * W[k * SR + lane];
    W[r * SR + lane] = v * rdp[r];
  }
  // back through the mean projection: t = L_old^-T z (in place)
  float gd = 0.f;
  for (int r = 0; r < K; ++r) gd += g * Y[r * SR + lane] * (MU[r * SR + lane] - MO[r * SR + lane]);
  if (active) {                                          // lanes of inactive rows skip the solve
    for (int r = K - 1; r >= 0; --r) {
      float v = Z[r * SR + lane];
      for (int k = r + 1; k < K; ++k) v -= Lo[k * LP + r] * Z[k * SR + lane];
      Z[r * SR + lane] = v * rdo[r];
    }
  }
  const float coef = active ? gd / (2.f * h.eps_mean * sc * sc * sc) : 0.f;
  const float trc = rok ? h.tr_coeff / (float)N : 0.f;
  // total gradient w.r.t. the mean net's output -> AC ([k][64]) and G3S (row-major)
  for (int r = 0; r < K; ++r) {
    const float gp = g * Y[r * SR + lane];
    float gm = active ? gp / sc - coef * Z[r * SR + lane] : gp;
    gm += trc * W[r * SR + lane];
    AC[r * SR + lane] = gm;
    S[L.g3s + lane * L.gp + r] = gm;
  }
  const double f = rok ? 1.0 : 0.0;
  sums[0] = wave_sum_f64(f * (double)ra);
  sums[1] = wave_sum_f64(f * (double)ratio);
  sums[2] = wave_sum_f64(f * (double)quad);
  sums[3] = wave_sum_f64(f * (double)maha2);
  sums[4] = wave_sum_f64(f * (double)quad / ((double)den * (double)den));
}

// The same head with the K-vectors in REGISTERS (K <= KP <= 32, loops unrolled
// over the padded length, rows / columns past K skipped by uniform branches):
// the [k][lane] LDS vectors above make every step of a triangular solve two
// dependent LDS round trips -- 60 of the row kernel's 80 us at K = 20.
// lower-triangular solve v <- M^-1 v (s_solve) / v <- M^-T v (s_solve_t) for a
// vector in registers; M: [KP][KP] LDS image (zero past K), rd: reciprocal
// diagonal (zero past K).  One row at a time: its entries are fetched
// (constant offsets), then the dot product; the scheduling barrier keeps the
// reads of ALL rows from being hoisted to the top.
// out[s] = T in[s] for NV vectors in registers and a TRIANGULAR matrix T given
// by the image R = T^T (LOWER: T lower triangular, so R[k][r] = T[r][k] is
// non-zero for r >= k) or, !LOWER, T upper triangular with R[k][r] = T[r][k]
// non-zero for r <= k: per input element k one contiguous piece of row k of
// the image (16-byte reads, constant offsets) and independent FMAs into all
// outputs -- no dependent chain, unlike a triangular solve, whose every row
// waits for the rows before it (6 solves of K = 20: 50 000 cycles per tile).
// The image is [KP][KP] with zeros past K; k >= K is skipped (uniform test,
// which also keeps each step a basic block of its own: one block of 24 x 24
// reads made the compiler hoist them all and spill).
template <int KP, int NV, bool LOWER>
__device__ __forceinline__ void s_trimv(float (&out)[NV][KP], const float (&in)[NV][KP],
                                        const float* R, int K) {
#pragma unroll
  for (int s = 0; s < NV; ++s)
#pragma unroll
    for (int r = 0; r < KP; ++r) out[s][r] = 0.f;
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    if (k < K) {
      constexpr int Q = KP / 4;
      const int q0 = LOWER ? k / 4 : 0, q1 = LOWER ? Q : k / 4 + 1;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        if (q >= q0 && q < q1) {                      // compile-time after unrolling
          const f32x4 t = *reinterpret_cast<const f32x4*>(R + k * KP + 4 * q);
#pragma unroll
          for (int s = 0; s < NV; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) out[s][4 * q + i] += t[i] * in[s][k];
        }
      }
    }
  }
}

// The black-box policy head with the K-vectors in registers (K <= KP <= 32)
// and the triangular solves as products with the explicit inverses of the two
// shared factors (Li_old = L_old^-1, Li_proj = L_proj^-1, computed once per
// update / once per epoch in double precision): six small dense layers.
// DG (diagonal factors, the reference's std_only configuration): every product
// with a factor's inverse is an elementwise product with the reciprocal
// diagonal (S + L.rdo / L.rdp, zero past K) -- the six dense K x K products
// were 22 800 of the row kernel's 81 000 cycles at K = 20.
template <int KP, bool DG>
__device__ __forceinline__ void s_diag_mv(float* out, const float* in, const float* rd) {
#pragma unroll
  for (int q = 0; q < KP; q += 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(rd + q);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[q + i] = t[i] * in[q + i];
  }
}
template <int KP, bool DG>
__device__ inline void s_policy_head_reg(const SLds& L, float* S, const SPolicyHead& h,
                                         int64_t r0, int64_t N, int K, int lane, float logdet_p,
                                         double (&sums)[5] SMLP_HARGS_DECL) {
  const int VS = L.doutp * SR;
  const float* MU = S + L.vec;
  const float* MO = MU + VS;
  float* AC = S + L.vec + 2 * VS;
  const int64_t row = r0 + lane;
  const bool rok = row < N;
  const int64_t rc = rok ? row : N - 1;
  const float lpo = h.logp_old[rc], adv = h.adv[rc];   // in flight during the first product
  float mu[KP], mo[KP], d[1][KP], z[1][KP], ab[2][KP], yw[2][KP], uq[2][KP];
#pragma unroll
  for (int r = 0; r < KP; ++r) {
    const int rr = r < K ? r : K - 1;
    const float live = r < K ? 1.f : 0.f;
    mu[r] = live * MU[rr * SR + lane];
    mo[r] = live * MO[rr * SR + lane];
    ab[0][r] = live * AC[rr * SR + lane];           // the sampled parameters
    d[0][r] = mu[r] - mo[r];
  }
  // z = L_old^-1 (mu - mu_old)
  SMLP_TH(1)
  if (DG) s_diag_mv<KP, DG>(z[0], d[0], S + L.rdo);
  else s_trimv<KP, 1, true>(z, d, S + L.liot, K);
  float quad = 0.f;
#pragma unroll
  for (int r = 0; r < KP; ++r) quad += z[0][r] * z[0][r];
  const float m = 0.5f * quad;
  const bool active = m > h.eps_mean;
  const float sc = active ? sqrtf(m / h.eps_mean) : 1.f;
  const float om = sc - 1.f;
  const float den = 1.f + om + 1e-16f;
  // y = L_proj^-1 (a - proj_mean); w = L_proj^-1 (mu - proj_mean)
#pragma unroll
  for (int r = 0; r < KP; ++r) {
    const float pm = active ? (mu[r] + om * mo[r]) / den : mu[r];
    if (r < K) {
      if (h.pmean_out && rok) h.pmean_out[row * K + r] = pm;
      if (h.mean_out && rok) h.mean_out[row * K + r] = mu[r];
    }
    ab[0][r] -= pm;
    ab[1][r] = mu[r] - pm;
  }
  if (DG) {
    s_diag_mv<KP, DG>(yw[0], ab[0], S + L.rdp);
    s_diag_mv<KP, DG>(yw[1], ab[1], S + L.rdp);
  } else {
    s_trimv<KP, 2, true>(yw, ab, S + L.lipt, K);
  }
  float quady = 0.f, maha2 = 0.f;
#pragma unroll
  for (int r = 0; r < KP; ++r) {
    quady += yw[0][r] * yw[0][r];
    maha2 += yw[1][r] * yw[1][r];
  }
  SMLP_TH(2)
  const float logp = -0.5f * quady - logdet_p - S_HALF_LOG_2PI * (float)K;
  const float ratio = expf(logp - lpo);
  const float ra = ratio * adv;
  const float g = rok ? -ra / (float)N : 0.f;
#pragma unroll
  for (int r = 0; r < KP; ++r)
    if (r < K) S[L.ys + lane * L.gp + r] = yw[0][r];
  // u = L_proj^-T y, q = L_proj^-T w; t = L_old^-T z
  if (DG) {
    s_diag_mv<KP, DG>(uq[0], yw[0], S + L.rdp);
    s_diag_mv<KP, DG>(uq[1], yw[1], S + L.rdp);
  } else {
    s_trimv<KP, 2, false>(uq, yw, S + L.lip, K);
  }
#pragma unroll
  for (int r = 0; r < KP; ++r)
    if (r < K) S[L.gus + lane * L.gp + r] = g * uq[0][r];
  float gd = 0.f;
#pragma unroll
  for (int r = 0; r < KP; ++r) gd += g * uq[0][r] * d[0][r];
  float t[1][KP];
#pragma unroll
  for (int r = 0; r < KP; ++r) t[0][r] = 0.f;
  if (DG) s_diag_mv<KP, DG>(t[0], z[0], S + L.rdo);
  else if (__builtin_amdgcn_ballot_w64(active) != 0)   // no row of the wave projected: skip
    s_trimv<KP, 1, false>(t, z, S + L.lio, K);
  const float coef = active ? gd / (2.f * h.eps_mean * sc * sc * sc) : 0.f;
  const float trc = rok ? h.tr_coeff / (float)N : 0.f;
#pragma unroll
  for (int r = 0; r < KP; ++r) {
    if (r < K) {
      const float gp = g * uq[0][r];
      float gm = active ? gp / sc - coef * t[0][r] : gp;
      gm += trc * uq[1][r];
      AC[r * SR + lane] = gm;
      S[L.g3s + lane * L.gp + r] = gm;
    }
  }
  SMLP_TH(3)
  const double f = rok ? 1.0 : 0.0;
  sums[0] = wave_sum_f64(f * (double)ra);
  sums[1] = wave_sum_f64(f * (double)ratio);
  sums[2] = wave_sum_f64(f * (double)quad);
  sums[3] = wave_sum_f64(f * (double)maha2);
  sums[4] = wave_sum_f64(f * (double)quad / ((double)den * (double)den));
  SMLP_TH(4)
}

// Diagonal factors, ALL FOUR waves: wave w takes rows 16 w .. 16 w + 15 of the
// tile, lane = (row, k-group kg = lane / 16) with the elements k = kg + 4 j,
// j < KP / 4; a sum over k is a sum over a lane's elements and over the four
// lanes of a row (two cross-lane steps).  One wave walking all K elements of
// its 64 rows was issue-bound: 11 600 of the row kernel's 56 000 cycles, with
// three waves waiting at the barrier.  Inputs: MU [k][64] (the forward pass),
// MO / AC row-major [64][K] and logp_old / adv in slot 4 (s_dma_tile); the
// gradient goes to slot 3 as [k][64] for the backward pass (not over AC: other
// waves still read their rows of it).  part: [SNW][5] doubles, the waves' sums.
template <int KP>
__device__ inline void s_policy_head_diag(const SLds& L, float* S, const SPolicyHead& h,
                                          int64_t r0, int64_t N, int K, int lane, int wave,
                                          float logdet_p, double* part) {
  constexpr int KQ = KP / 4;
  const int VS = L.doutp * SR;
  const float* MU = S + L.vec;
  const float* MO = MU + VS;
  const float* AC = MU + 2 * VS;
  float* G = S + L.vec + 3 * VS;
  const int rt = 16 * wave + (lane & 15), kg = lane >> 4;
  const int64_t row = r0 + rt;
  const bool rok = row < N;
  const float lpo = S[L.vec + 4 * VS + rt], adv = S[L.vec + 4 * VS + SR + rt];
  auto ksum = [](float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
  };
  float mu[KQ], mo[KQ], ac[KQ], ro[KQ], rp[KQ], d[KQ], z[KQ];
  bool live[KQ];
  float quad = 0.f;
#pragma unroll
  for (int j = 0; j < KQ; ++j) {
    const int k = kg + 4 * j;
    live[j] = k < K;
    const int kk = live[j] ? k : K - 1;
    const float lv = live[j] ? 1.f : 0.f;
    mu[j] = lv * MU[kk * SR + rt];
    mo[j] = lv * MO[rt * K + kk];
    ac[j] = lv * AC[rt * K + kk];
    ro[j] = S[L.rdo + k];                           // (zero past K)
    rp[j] = S[L.rdp + k];
    d[j] = mu[j] - mo[j];
    z[j] = d[j] * ro[j];                            // z = L_old^-1 (mu - mu_old)
    quad += z[j] * z[j];
  }
  quad = ksum(quad);
  const float m = 0.5f * quad;
  const bool active = m > h.eps_mean;
  const float sc = active ? sqrtf(m / h.eps_mean) : 1.f;
  const float om = sc - 1.f;
  const float den = 1.f + om + 1e-16f;
  float y[KQ], w[KQ];
  float quady = 0.f, maha2 = 0.f;
#pragma unroll
  for (int j = 0; j < KQ; ++j) {
    const int k = kg + 4 * j;
    const float pm = active ? (mu[j] + om * mo[j]) / den : mu[j];
    if (live[j] && rok) {
      if (h.pmean_out) h.pmean_out[row * K + k] = pm;
      if (h.mean_out) h.mean_out[row * K + k] = mu[j];
    }
    y[j] = (ac[j] - pm) * rp[j];                    // y = L_proj^-1 (a - proj_mean)
    w[j] = (mu[j] - pm) * rp[j];                    // w = L_proj^-1 (mu - proj_mean)
    quady += y[j] * y[j];
    maha2 += w[j] * w[j];
  }
  quady = ksum(quady);
  maha2 = ksum(maha2);
  const float logp = -0.5f * quady - logdet_p - S_HALF_LOG_2PI * (float)K;
  const float ratio = expf(logp - lpo);
  const float ra = ratio * adv;
  const float g = rok ? -ra / (float)N : 0.f;
  float gu[KQ];
  float gd = 0.f;
#pragma unroll
  for (int j = 0; j < KQ; ++j) {
    const int k = kg + 4 * j;
    gu[j] = g * (y[j] * rp[j]);                     // g u, u = L_proj^-T y
    gd += gu[j] * d[j];
    if (live[j]) {
      S[L.ys + rt * L.gp + k] = y[j];
      S[L.gus + rt * L.gp + k] = gu[j];
    }
  }
  gd = ksum(gd);
  const float coef = active ? gd / (2.f * h.eps_mean * sc * sc * sc) : 0.f;
  const float trc = rok ? h.tr_coeff / (float)N : 0.f;
#pragma unroll
  for (int j = 0; j < KQ; ++j) {
    const int k = kg + 4 * j;
    if (live[j]) {
      float gm = active ? gu[j] / sc - coef * (z[j] * ro[j]) : gu[j];   // t = L_old^-T z
      gm += trc * (w[j] * rp[j]);                                       // q = L_proj^-T w
      G[k * SR + rt] = gm;
      S[L.g3s + rt * L.gp + k] = gm;
    }
  }
  const double f = rok && kg == 0 ? 1.0 : 0.0;
  const double s0 = wave_sum_f64(f * (double)ra);
  const double s1 = wave_sum_f64(f * (double)ratio);
  const double s2 = wave_sum_f64(f * (double)quad);
  const double s3 = wave_sum_f64(f * (double)maha2);
  const double s4 = wave_sum_f64(f * (double)quad / ((double)den * (double)den));
  if (lane == 0) {
    double* p = part + 5 * wave;
    p[0] = s0; p[1] = s1; p[2] = s2; p[3] = s3; p[4] = s4;
  }
}

// ---------------------------------------------------------------------------
template <int H, int ACT>
__global__ __launch_bounds__(SBT) void smlp_forward_kernel(SNet n, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float S[];
  const SLds L = s_lds(n.din, H, n.dout, HEAD_NONE);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* ov = S + L.vec;                 // needs doutp * 64 floats: see s_forward_lds
  s_load_weights<H>(L, S, n);
  for (int64_t r0 = (int64_t)blockIdx.x * SR; r0 < n.N; r0 += (int64_t)gridDim.x * SR) {
    __syncthreads();
    s_load_x(L, S, n, r0);
    __syncthreads();
    float h1[H / SNW], h2[H / SNW];
    s_forward<H, ACT>(L, S, n.dout, lane, wave, h1, h2, ov, n.act);
    for (int e = threadIdx.x; e < SR * n.dout; e += SBT) {
      const int r = e / n.dout, o = e - r * n.dout;
      if (r0 + r < n.N) out[(r0 + r) * n.dout + o] = ov[o * SR + r];
    }
  }
}
// the forward kernel keeps its outputs where the heads keep their vectors
__host__ __device__ inline int s_forward_lds_floats(int din, int H, int dout) {
  const SLds L = s_lds(din, H, dout, HEAD_NONE);
  return L.vec + s_up4(dout) * SR + 16;
}

template <int H, int ACT, int HEAD, int KP, bool DG>
__global__ __launch_bounds__(SBT) void smlp_epoch_kernel(SNet n, SValueHead vh, SPolicyHead ph,
                                                        SReduce rd) {
  extern __shared__ __attribute__((aligned(16))) float S[];
  const SLds L = s_lds(n.din, H, n.dout, HEAD);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = n.dout;
  constexpr bool POL = HEAD == HEAD_BB_POLICY;
  SMLP_T0()
  // weights, the first tile and (diagonal factors) the two diagonals: one round trip
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  s_dma_weights<H>(L, S, n, wave_u, (unsigned)lane);
  s_dma_tile<POL, POL && DG>(L, S, n, ph, (int64_t)blockIdx.x * SR, wave_u, (unsigned)lane);
  float logdet_p = 0.f;
  float dol = 1.f, dpl = 1.f;
  if (POL && DG && lane < K) {                         // (every wave runs the head)
    dol = ph.L_old[lane * K + lane];
    dpl = ph.L_proj[lane * K + lane];
  }
  if (POL && DG) {
    if (tid < KP) {
      S[L.rdo + tid] = tid < K ? 1.f / dol : 0.f;
      S[L.rdp + tid] = tid < K ? 1.f / dpl : 0.f;
    }
    logdet_p = wave_sum(lane < K ? logf(dpl) : 0.f);
  }
  if (POL && !DG) {
    const int LP = s_kpad(K);
    for (int e = tid; e < LP * LP; e += SBT) {
      const int i = e / LP, j = e - i * LP;
      const bool in = i < K && j < K;
      S[L.lo + e] = in ? ph.L_old[i * K + j] : 0.f;
      S[L.lp + e] = in ? ph.L_proj[i * K + j] : 0.f;
    }
    for (int e = tid; e < L.doutp; e += SBT) {
      S[L.rdo + e] = e < K ? 1.f / ph.L_old[e * K + e] : 0.f;
      S[L.rdp + e] = e < K ? 1.f / ph.L_proj[e * K + e] : 0.f;
    }
    if (KP > 0) {
      for (int e = tid; e < LP * LP; e += SBT) {
        const int i = e / LP, j = e - i * LP;
        const bool in = i < K && j < K;
        const float a = in ? ph.Li_old[i * K + j] : 0.f, b = in ? ph.Li_proj[i * K + j] : 0.f;
        S[L.lio + e] = a;
        S[L.liot + j * LP + i] = a;
        S[L.lip + e] = b;
        S[L.lipt + j * LP + i] = b;
      }
    }
    logdet_p = wave_sum(lane < K ? logf(ph.L_proj[lane * K + lane]) : 0.f);
  }
  float* slab = rd.slabs + (int64_t)blockIdx.x * rd.PS;
  double acc_d[5] = {0, 0, 0, 0, 0};
  bool first = true;
  float* vecs = S + L.vec;
  s_dma_wait();
  s_zero_w1_pad<H>(L, S, n);
  for (int64_t r0 = (int64_t)blockIdx.x * SR; r0 < n.N; r0 += (int64_t)gridDim.x * SR) {
    if (!first) {
      __syncthreads();
      s_dma_tile<POL, POL && DG>(L, S, n, ph, r0, wave_u, (unsigned)lane);
      s_dma_wait();
    }
    __syncthreads();
    float h1[H / SNW], h2[H / SNW];
    SMLP_T(0)
    s_forward<H, ACT>(L, S, K, lane, wave, h1, h2, vecs, n.act);
    SMLP_TK(1)
    SMLP_TH(0)
    const float* g;
    if (HEAD == HEAD_VALUE) {
      if (wave == 0) acc_d[0] += s_value_head(L, S, vh, r0, n.N, lane, vecs, vecs + SR);
      g = vecs + SR;
    } else if (DG) {
      // (the K x K image slots are free with diagonal factors: the waves' sums)
      double* part = reinterpret_cast<double*>(S + L.lo);
      s_policy_head_diag<(KP > 0 ? KP : 4)>(L, S, ph, r0, n.N, K, lane, wave, logdet_p, part);
      g = vecs + 3 * L.doutp * SR;
    } else {
      if (wave == 0) {
        double s5[5];
        if (KP > 0)
          s_policy_head_reg<(KP > 0 ? KP : 4), false>(L, S, ph, r0, n.N, K, lane, logdet_p,
                                                      s5 SMLP_HARGS);
        else s_policy_head(L, S, ph, r0, n.N, K, lane, logdet_p, s5);
#pragma unroll
        for (int i = 0; i < 5; ++i) acc_d[i] += s5[i];
      }
      g = vecs + 2 * L.doutp * SR;                     // AC slot
    }
    __syncthreads();
    if (POL && DG && tid == 0) {
      const double* part = reinterpret_cast<const double*>(S + L.lo);
#pragma unroll
      for (int w = 0; w < SNW; ++w)
#pragma unroll
        for (int i = 0; i < 5; ++i) acc_d[i] += part[5 * w + i];
    }
    SMLP_TK(2)
    s_backward<H, ACT>(L, S, K, lane, wave, h1, h2, g, n.act);
    SMLP_TK(3)
#ifdef SMLP_GRADS_VALU
    s_param_grads<H>(L, S, n.din, K, slab, !first,
                     HEAD == HEAD_BB_POLICY ? slab + rd.P : nullptr);
#else
    s_param_grads_mfma<H>(L, S, n.din, K, slab, !first,
                          HEAD == HEAD_BB_POLICY ? slab + rd.P : nullptr, wave_u, lane);
#endif
    first = false;
    SMLP_TK(4)
  }
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < 5; ++i) rd.dpart[blockIdx.x * 8 + i] = acc_d[i];
  }
#ifdef SMLP_STAMP
  // diagnostic build (scripts/smlp_stamps.py): cycles of workgroup 0, wave 0 --
  // {setup + loads, forward, head, backward, gradients} behind the slabs
  if (blockIdx.x == 0 && tid == 0)
    for (int i = 0; i < 5; ++i)       // the 8 spare floats at the end of the workspace
      rd.slabs[(int64_t)gridDim.x * rd.PS + 16 * (int64_t)gridDim.x + 24 + i] = (float)stamp_[i];
#endif
}

// ---------------------------------------------------------------------------
// column sums of the slabs in a fixed order: FIN_GROUPS interleaved groups of
// slabs in parallel, then the groups (as mlp_finish_kernel).  Columns [0, P):
// the network's gradient (+ the Adam update of torch.optim.Adam when f.param);
// columns [P, P + KK): f.extra.  Workgroup 0 also adds the per-workgroup
// double sums.
__global__ __launch_bounds__(64 * FIN_GROUPS) void smlp_reduce_kernel(SFinish f) {
  __shared__ float part[FIN_GROUPS][64];
  __shared__ float red[FIN_GROUPS];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int p = blockIdx.x * 64 + col;
  const int tot = f.P + f.KK;
  float s = 0.f;
  if (p < tot) {
    const float* src = f.slabs + p;
#pragma unroll 4
    for (int i = grp; i < f.nparts; i += FIN_GROUPS) s += src[(int64_t)i * f.PS];
  }
  part[grp][col] = s;
  __syncthreads();
  float sq = 0.f;
  if (grp == 0 && p < tot) {
    float g0 = 0.f;
#pragma unroll
    for (int k = 0; k < FIN_GROUPS; ++k) g0 += part[k][col];   // fixed order
    if (p < f.P) {
      f.grad[p] = g0;
      sq = g0 * g0;
      if (f.param) {
        float w = f.param[p], mi = f.m[p], vi = f.v[p], step_size, bc2s;
        adam_coef(f.lr, f.b1, f.b2, f.step, step_size, bc2s);
        adam_elem(g0 * f.gscale, w, mi, vi, f.b1, f.b2, f.eps, f.wd, step_size, bc2s);
        f.m[p] = mi;
        f.v[p] = vi;
        f.param[p] = w;
      }
    } else {
      f.extra[p - f.P] = g0;
    }
  }
  const float tot_sq = block_sum(sq, red);
  if (threadIdx.x == 0 && f.stats) atomicAdd(&f.stats[1], tot_sq);
  if (blockIdx.x == 0) {
    // per-workgroup double sums, 8 columns, slab order
    __shared__ double dsh[FIN_GROUPS][8];
    const int c = threadIdx.x & 7, gq = threadIdx.x >> 3;        // 128 groups of 8 columns
    double d = 0;
    if (gq < FIN_GROUPS)
      for (int i = gq; i < f.nparts; i += FIN_GROUPS) d += f.dpart[i * 8 + c];
    if (gq < FIN_GROUPS) dsh[gq][c] = d;
    __syncthreads();
    if (threadIdx.x < 8) {
      double t = 0;
#pragma unroll
      for (int k = 0; k < FIN_GROUPS; ++k) t += dsh[k][threadIdx.x];
      f.dsum[threadIdx.x] = t;
      if (threadIdx.x == 0 && f.stats) f.stats[0] = (float)(t / (double)f.N);
      if (threadIdx.x == 0 && f.param) f.state[0] = f.step;
    }
  }
}

// Li = L^-1 for a lower-triangular [K,K] factor (one workgroup, double
// precision in LDS): the row kernel applies the inverses as dense products
__global__ __launch_bounds__(SM_BT) void tri_inverse_kernel(const float* __restrict__ Lm,
                                                            float* __restrict__ Li, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int KP = sm_pitch(K);
  double* A = reinterpret_cast<double*>(smem_raw);
  double* X = A + K * KP;
  sm_load(A, Lm, K, KP, true);
  for (int e = threadIdx.x; e < K * K; e += SM_BT) {
    const int i = e / K, j = e - i * K;
    X[i * KP + j] = i == j ? 1.0 : 0.0;
  }
  __syncthreads();
  sm_trsm_l(X, A, K, KP);
  for (int e = threadIdx.x; e < K * K; e += SM_BT) {
    const int i = e / K, j = e - i * K;
    Li[e] = j <= i ? (float)X[i * KP + j] : 0.f;
  }
}
inline int s_tri_inverse(const float* Lm, float* Li, int K, hipStream_t st) {
  const size_t lds = 2 * (size_t)K * sm_pitch(K) * sizeof(double);
  tce_lds_limit(reinterpret_cast<const void*>(tri_inverse_kernel), lds);
  hipLaunchKernelGGL(tri_inverse_kernel, dim3(1), dim3(SM_BT), lds, st, Lm, Li, K);
  TCE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------
// Matrix side of a black-box policy epoch for DIAGONAL factors (std_only):
// Cholesky head, KL covariance projection (oracle/kl_oracle.py cov_projection
// restricted to diagonal matrices: lambda_k = (sigma_k / sigma_old_k)^2, eta
// the root of 1/2 sum (mu_k - 1 - log mu_k) = eps, mu_k = (eta + 1) lambda_k /
// (eta lambda_k + 1)), the covariance halves of the three KL terms, the
// entropy of the projected policy and the trust region gradient w.r.t. the new
// factor.  One wave.  dctx (double): [lambda K | mu K | eta, active, -, -].
// (A device function: its own 64-thread kernel before the first epoch; behind
// an epoch's Adam step the finish kernel runs it for the NEXT epoch -- one launch
// less in a chain of dependent launches.  Called by whole workgroups; wave 0
// does the work, `dg` is the caller's [3][64] LDS array.)
__device__ inline void bb_diag_fwd_body(
    const float* __restrict__ var, float min_std, const float* __restrict__ L_old, int K,
    double eps_cov, float tr_coeff, int include_cov, float* __restrict__ L_new,
    float* __restrict__ L_proj, float* __restrict__ Li_proj, float* __restrict__ gL_tr,
    float* __restrict__ out16, double* __restrict__ dctx, float (*dg)[64],
    float* __restrict__ kl_rec = nullptr) {
  if (threadIdx.x >= 64) {
    __syncthreads();                                     // (the barrier below)
    return;
  }
  const int k = threadIdx.x;
  const bool live = k < K;
  double sig = 1, so = 1;
  if (live) {
    const float x = var[k];
    sig = (double)((x > 20.f ? x : log1pf(expf(x))) + min_std);
    so = (double)L_old[k * K + k];
  }
  const double a = sig / so, lam = a * a;
  const double kl0 = 0.5 * wave_sum_f64(live ? lam - 1.0 - log(lam) : 0.0);
  const bool active = kl0 > eps_cov;
  double eta = 0, mu = lam;
  if (active) {
    auto hfun = [&](double e) {
      double t = 0;
      if (live) {
        const double m_ = (e + 1.0) * lam / (e * lam + 1.0);
        t = m_ - 1.0 - log(m_);
      }
      return 0.5 * wave_sum_f64(t);
    };
    double lo = 0.0, hi = 1.0;
    for (int i = 0; i < 200 && hfun(hi) > eps_cov; ++i) { lo = hi; hi *= 2.0; }
    double eta_n = lo;
    for (int i = 0; i < 60; ++i) {
      double t = 0, dt = 0;
      if (live) {
        const double w = 1.0 / (eta_n * lam + 1.0);
        const double m_ = (eta_n + 1.0) * lam * w;
        t = m_ - 1.0 - log(m_);
        dt = (1.0 - 1.0 / m_) * lam * (1.0 - lam) * w * w;
      }
      const double hv = 0.5 * wave_sum_f64(t) - eps_cov;
      const double dh = 0.5 * wave_sum_f64(dt);
      if (hv > 0) lo = eta_n; else hi = eta_n;
      double nxt = eta_n - hv / dh;
      if (!(nxt > lo && nxt < hi)) nxt = 0.5 * (lo + hi);
      if (fabs(nxt - eta_n) <= 1e-15 * fabs(nxt)) { eta_n = nxt; break; }
      eta_n = nxt;
    }
    eta = eta_n;
    mu = (eta + 1.0) * lam / (eta * lam + 1.0);
  }
  const double pl = active ? so * sqrt(mu) : sig;
  // the matrices (diagonal; everything else zero)
  dg[0][k] = (float)sig;
  dg[1][k] = (float)pl;
  dg[2][k] = include_cov ? (float)((double)tr_coeff * (sig / (pl * pl) - 1.0 / sig)) : 0.f;
  __syncthreads();
  for (int e = k; e < K * K; e += 64) {
    const int i = e / K, j = e - i * K;
    L_new[e] = i == j ? dg[0][i] : 0.f;
    L_proj[e] = i == j ? dg[1][i] : 0.f;
    Li_proj[e] = i == j ? 1.f / dg[1][i] : 0.f;
    gL_tr[e] = i == j ? dg[2][i] : 0.f;
  }
  if (live) {
    dctx[k] = lam;
    dctx[K + k] = mu;
  }
  // covariance halves of (new || old), (new || proj), (proj || old)
  const double r0 = sig / so, r1 = sig / pl, r2 = pl / so;
  const double f0 = wave_sum_f64(live ? r0 * r0 : 0.0), l0 = wave_sum_f64(live ? log(r0) : 0.0);
  const double f1 = wave_sum_f64(live ? r1 * r1 : 0.0), l1 = wave_sum_f64(live ? log(r1) : 0.0);
  const double f2 = wave_sum_f64(live ? r2 * r2 : 0.0), l2 = wave_sum_f64(live ? log(r2) : 0.0);
  const double lpld = wave_sum_f64(live ? log(pl) : 0.0);
  if (k == 0) {
    const double f[3] = {f0, f1, f2}, l[3] = {l0, l1, l2};
    for (int q = 0; q < 3; ++q) {
      const double shape = 0.5 * (f[q] - (double)K), volume = -l[q];
      out16[4 * q + 0] = 0.f;                            // mean part: the finish kernel
      out16[4 * q + 1] = (float)(shape + volume);
      out16[4 * q + 2] = (float)shape;
      out16[4 * q + 3] = (float)volume;
      if (kl_rec) {                                      // the epoch's record row (KL means)
        kl_rec[4 * q + 1] = (float)(shape + volume);
        kl_rec[4 * q + 2] = (float)shape;
        kl_rec[4 * q + 3] = (float)volume;
      }
    }
    out16[12] = (float)(0.5 * (double)K * (1.0 + 1.8378770664093453) + lpld);
    out16[13] = 0.f;
    out16[14] = 0.f;
    out16[15] = 0.f;
    dctx[2 * K] = eta;
    dctx[2 * K + 1] = active ? 1.0 : 0.0;
  }
}

__global__ __launch_bounds__(64) void bb_diag_fwd_kernel(
    const float* __restrict__ var, float min_std, const float* __restrict__ L_old, int K,
    double eps_cov, float tr_coeff, int include_cov, float* __restrict__ L_new,
    float* __restrict__ L_proj, float* __restrict__ Li_proj, float* __restrict__ gL_tr,
    float* __restrict__ out16, double* __restrict__ dctx, float* __restrict__ kl_rec) {
  __shared__ float dg[3][64];
  bb_diag_fwd_body(var, min_std, L_old, K, eps_cov, tr_coeff, include_cov, L_new, L_proj, Li_proj,
                   gL_tr, out16, dctx, dg, kl_rec);
}

// finish of a black-box policy epoch (one workgroup): the mean parts of the KL
// terms and the trust region loss from the Mahalanobis sums, the corrections
// on the diagonal of d / d L_proj, -- diagonal factors -- the covariance
// projection's backward, g_L = dTR/dL_new + projection backward, Cholesky head
// backward into the variance slots of the flat gradient, global norm, clip
// factor, Adam, record row {surrogate, entropy loss, trust region loss, total,
// entropy, |g|, |g| clipped}.
// phase 0 (general factors): only the corrections of g_pL (before
//   tce_kl_cov_proj_bwd); phase 1: everything else; diag != 0: both at once.
__global__ __launch_bounds__(SBT) void bb_policy_finish_kernel(
    int phase, int diag, float* __restrict__ g_pL, const float* __restrict__ gL_tr,
    const float* __restrict__ gL_p, const float* __restrict__ L_proj,
    const float* __restrict__ L_old, const double* __restrict__ dctx,
    const double* __restrict__ dsum, int64_t N, int K, int nvec, int P, float tr_coeff,
    int include_cov, float ent_coef, float* __restrict__ param, float* __restrict__ grad,
    float* __restrict__ m, float* __restrict__ v, float* __restrict__ state, float lr, float b1,
    float b2, float eps, float wd, float clip_grad, float gscale, int do_adam,
    float* __restrict__ out16, float* __restrict__ rec, int rec_kl,
    // next_fwd (diagonal factors, behind the Adam step): bb_diag_fwd_body of the NEXT epoch
    int next_fwd, float min_std, double eps_cov, float* __restrict__ L_new_w,
    float* __restrict__ L_proj_w, float* __restrict__ Li_proj_w, float* __restrict__ gL_tr_w,
    double* __restrict__ dctx_w, XchgView X) {
  __shared__ float red[SNW];
  __shared__ float coef_s, step_s;
  __shared__ double gsh[64];
  __shared__ float dg[3][64];
  const int tid = threadIdx.x;
  const double invN = 1.0 / (double)N;
  const float sum_g = (float)(-dsum[0] * invN);          // sum over rows of d surrogate / d logp
  if (phase == 0 || diag) {
    for (int e = tid; e < K * K; e += SBT) {
      const int i = e / K, j = e - i * K;
      float x = j <= i ? g_pL[e] : 0.f;
      if (i == j) x -= (sum_g + ent_coef) / L_proj[e];
      g_pL[e] = x;
    }
    if (phase == 0 && !diag) return;
    __syncthreads();
  }
  const float* var = param + P;
  if (diag) {
    // covariance projection backward on the diagonal
    const double eta = dctx[2 * K];
    const bool active = dctx[2 * K + 1] != 0.0;
    if (tid < 64) {                                      // wave 0, all lanes take part in the sums
      const int k = tid;
      const bool live = k < K;
      double gp = 0, gmu = 0, dl = 0, de = 0, hm = 0, scale = 0;
      if (live) {
        gp = (double)g_pL[k * K + k];
        if (active) {
          const double lam = dctx[k], mu = dctx[K + k];
          const double so = (double)L_old[k * K + k], pl = (double)L_proj[k * K + k];
          const double w = 1.0 / (eta * lam + 1.0);
          gmu = gp * pl / (2.0 * mu);                    // d / d mu_k
          dl = (eta + 1.0) * w * w;                      // d mu_k / d lambda_k
          de = lam * (1.0 - lam) * w * w;                // d mu_k / d eta
          hm = 0.5 * (1.0 - 1.0 / mu);                   // d h / d mu_k
          scale = 2.0 * sqrt(lam) / so;                  // d lambda_k / d sigma_k
        }
      }
      const double h_eta = wave_sum_f64(hm * de);
      const double gde = wave_sum_f64(gmu * de);
      double gs = gp;
      if (active) gs = (gmu * dl - gde * (hm * dl) / h_eta) * scale;
      gsh[k] = live ? gs : 0.0;
    }
    __syncthreads();
  }
  for (int i = tid; i < nvec; i += SBT) {
    float gv;
    if (i < K) {
      const float x = var[i];
      const float sig = x > 20.f ? 1.f : 1.f / (1.f + expf(-x));
      const float gl = diag ? (float)gsh[i] : gL_p[i * K + i];
      gv = (gL_tr[i * K + i] + gl) * sig;
    } else {
      const int t = i - K;
      int r = (int)((1.0 + sqrt(1.0 + 8.0 * (double)t)) * 0.5);
      while (r * (r - 1) / 2 > t) --r;
      while ((r + 1) * r / 2 <= t) ++r;
      const int c = t - r * (r - 1) / 2;
      gv = gL_tr[r * K + c] + gL_p[r * K + c];
    }
    grad[P + i] = gv;
  }
  __syncthreads();
  const int n = P + nvec;
  if (xchg_on(X) && do_adam) {
    // env shards: the peers' gradients in rank order (one workgroup, one flag)
    for (int p = tid; p < n; p += SBT) xchg_put<float>(X, p, grad[p]);
    xchg_sync(X, 0);
    for (int p = tid; p < n; p += SBT) grad[p] = xchg_get<float>(X, p, grad[p]);
    __syncthreads();
  }
  float sq = 0.f;
  for (int p = tid; p < n; p += SBT) sq += grad[p] * grad[p];
  sq = block_sum(sq, red);
  if (tid == 0) {
    const float before = sqrtf(sq) * gscale;
    float coef = 1.f;
    if (clip_grad > 0.f) coef = fminf(clip_grad / (before + 1e-6f), 1.f);
    if (do_adam) {
      const float step = state[0] + 1.f;
      state[0] = step;
      state[1] = before;
      state[2] = before * coef;
      state[3] = coef * gscale;
      step_s = step;
    }
    coef_s = coef * gscale;
    // mean parts of the KL terms (new || old), (new || proj), (proj || old)
    double mp[3];
    for (int q = 0; q < 3; ++q) {
      mp[q] = 0.5 * dsum[2 + q] * invN;
      out16[4 * q] = (float)mp[q];
    }
    const double tr = (double)tr_coeff * (mp[1] + (include_cov ? (double)out16[5] : 0.0));
    out16[13] = (float)tr;
    const float sur = (float)(-dsum[0] * invN);
    const float entl = ent_coef == 0.f ? 0.f : -ent_coef * out16[12];
    rec[0] = sur;
    rec[1] = entl;
    rec[2] = (float)tr;
    rec[3] = ent_coef == 0.f ? sur + (float)tr : sur + (float)tr + entl;
    rec[4] = out16[12];
    rec[5] = before;
    rec[6] = before * coef;
    // kl_old_new_proj (black_box_agent.py:391-436): {mean, cov, shape, volume}
    // of (new || old), (new || proj), (proj || old) -- this epoch's K x K parts
    // are in out16 since the launch that built the factors
    if (rec_kl)
      for (int i = 0; i < 12; ++i) rec[7 + i] = out16[i];
  }
  __syncthreads();
  if (!do_adam) return;
  const float step = step_s, coef = coef_s;
  float step_size, bc2s;
  adam_coef(lr, b1, b2, step, step_size, bc2s);
  for (int p = tid; p < n; p += SBT) {
    float w = param[p], mi = m[p], vi = v[p];
    adam_elem(grad[p] * coef, w, mi, vi, b1, b2, eps, wd, step_size, bc2s);
    m[p] = mi;
    v[p] = vi;
    param[p] = w;
  }
  if (next_fwd) {
    // the next epoch's head / projection from the parameters just written (same
    // workgroup: visible behind the barrier)
    __threadfence_block();
    __syncthreads();
    bb_diag_fwd_body(param + P, min_std, L_old, K, eps_cov, tr_coeff, include_cov, L_new_w,
                     L_proj_w, Li_proj_w, gL_tr_w, out16, dctx_w, dg);
  }
}

// The finish of an epoch with DIAGONAL factors and the Adam step, written for
// the length of its dependent chain (bb_policy_finish_kernel above is general
// and, at 21 us per call, was the second longest link of a black-box epoch: six
// barrier-separated passes, each a round trip to memory, and a bracket + Newton
// search of ~10 double-precision steps):
//  * everything the kernel reads -- the gradient, the parameters, both
//    moments, the K-vectors -- is requested at the top: ONE round trip;
//  * the gradient and the parameters stay in registers from the norm to the
//    Adam step; the new variance parameters go from the Adam step to the next
//    epoch's projection in wave 0's registers;
//  * the multiplier eta of the covariance projection by Newton steps on
//    h(eta)^-1/2 - eps^-1/2 (nearly linear in eta: 2 - 4 steps from eta = 0,
//    no bracket search), safeguarded by the interval the signs have shown;
//  * the next epoch's matrices: only their diagonals are rewritten (the first
//    launch of the update, bb_diag_fwd_kernel, has zeroed the rest).
// P <= 256 * FD_NPT.  Same record row, state vector and out16 as the
// general kernel.
// (NPT: parameters per thread.  The loads are unconditional with clamped
// addresses and the values selected afterwards: element assignments under a
// branch make the compiler copy the whole register array at every step.)
// rec_kl: 0, or the record's row stride when the rows carry the 12 KL means at
// [7, 19) (the mean parts are written here, the K-vector parts of the NEXT
// epoch by the next_fwd section into the next row).
constexpr int FD_NPT = 16;
template <int NPT>
__global__ __launch_bounds__(SBT) void bb_diag_finish_kernel(
    const float* __restrict__ g_pL, const float* __restrict__ gL_tr,
    const float* __restrict__ L_proj, const float* __restrict__ L_old,
    const double* __restrict__ dctx, const double* __restrict__ dsum, int64_t N, int K, int P,
    float tr_coeff, int include_cov, float ent_coef, float* __restrict__ param,
    float* __restrict__ grad, float* __restrict__ m, float* __restrict__ v,
    float* __restrict__ state, float lr, float b1, float b2, float eps, float wd, float clip_grad,
    float gscale, float* __restrict__ out16, float* __restrict__ rec, int rec_kl, int next_fwd,
    float min_std, double eps_cov, float* __restrict__ L_new_w, float* __restrict__ L_proj_w,
    float* __restrict__ Li_proj_w, float* __restrict__ gL_tr_w, double* __restrict__ dctx_w,
    XchgView X) {
  __shared__ float red[SNW];
  __shared__ float coef_s;
  const int tid = threadIdx.x;
  // ---- requests
  float gr[NPT], pw[NPT], pm[NPT], pv[NPT];
#pragma unroll
  for (int q = 0; q < NPT; ++q) {
    const int p = tid + q * SBT, pc = p < P ? p : P - 1;
    const float g0 = grad[pc];
    gr[q] = p < P ? g0 : 0.f;
    pw[q] = param[pc];
    pm[q] = m[pc];
    pv[q] = v[pc];
  }
  const bool w0 = tid < 64, live = tid < K;
  const int kk = live ? tid * K + tid : 0;
  float gp_raw = 0.f, gtr = 0.f, lpj = 1.f, so_f = 1.f, var = 0.f, mv = 0.f, vv = 0.f;
  double lam = 1, mu = 1, eta = 0, act_d = 0, ds0 = 0, ds2 = 0, ds3 = 0, ds4 = 0;
  float o5 = 0.f, o12 = 0.f;
  const float step = state[0] + 1.f;
  if (w0) {
    if (live) {
      gp_raw = g_pL[kk];
      gtr = gL_tr[kk];
      lpj = L_proj[kk];
      so_f = L_old[kk];
      lam = dctx[tid];
      mu = dctx[K + tid];
      var = param[P + tid];
      mv = m[P + tid];
      vv = v[P + tid];
    }
    eta = dctx[2 * K];
    act_d = dctx[2 * K + 1];
    ds0 = dsum[0]; ds2 = dsum[2]; ds3 = dsum[3]; ds4 = dsum[4];
    o5 = out16[5];
    o12 = out16[12];
  }
  __builtin_amdgcn_sched_barrier(0);
  const double invN = 1.0 / (double)N;
  // ---- wave 0: d / d L_proj on the diagonal, projection backward, Cholesky head backward
  float gvar = 0.f;
  if (w0) {
    const float sum_g = (float)(-ds0 * invN);
    const bool active = act_d != 0.0;
    double gpd = 0, gmu = 0, dl = 0, de = 0, hm = 0, scale = 0;
    if (live) {
      gpd = (double)(gp_raw - (sum_g + ent_coef) / lpj);
      if (active) {
        const double so = (double)so_f, pl = (double)lpj;
        const double w = 1.0 / (eta * lam + 1.0);
        gmu = gpd * pl / (2.0 * mu);
        dl = (eta + 1.0) * w * w;
        de = lam * (1.0 - lam) * w * w;
        hm = 0.5 * (1.0 - 1.0 / mu);
        scale = 2.0 * sqrt(lam) / so;
      }
    }
    const double h_eta = wave_sum_f64(hm * de);
    const double gde = wave_sum_f64(gmu * de);
    double gs = gpd;
    if (active) gs = (gmu * dl - gde * (hm * dl) / h_eta) * scale;
    if (live) {
      const float sig = var > 20.f ? 1.f : 1.f / (1.f + expf(-var));
      gvar = (gtr + (float)gs) * sig;
      grad[P + tid] = gvar;
    }
  }
  // ---- env shards: the peers' gradients in rank order (one workgroup, one flag)
  if (xchg_on(X)) {
#pragma unroll
    for (int q = 0; q < NPT; ++q) {
      const int p = tid + q * SBT;
      if (p < P) xchg_put<float>(X, p, gr[q]);
    }
    if (w0 && live) xchg_put<float>(X, P + tid, gvar);
    xchg_sync(X, 0);
#pragma unroll
    for (int q = 0; q < NPT; ++q) {
      const int p = tid + q * SBT, pc = p < P ? p : P - 1;
      const float sum = xchg_get<float>(X, pc, gr[q]);
      gr[q] = p < P ? sum : 0.f;
      if (p < P) grad[p] = sum;
    }
    if (w0 && live) {
      gvar = xchg_get<float>(X, P + tid, gvar);
      grad[P + tid] = gvar;
    }
  }
  // ---- global norm, clip factor
  float sq = gvar * gvar;
#pragma unroll
  for (int q = 0; q < NPT; ++q) sq += gr[q] * gr[q];
  sq = block_sum(sq, red);
  const float before = sqrtf(sq) * gscale;
  float cf = 1.f;
  if (clip_grad > 0.f) cf = fminf(clip_grad / (before + 1e-6f), 1.f);
  const float coef = cf * gscale;
  if (tid == 0) {
    state[0] = step;
    state[1] = before;
    state[2] = before * cf;
    state[3] = coef;
    double mp1 = 0;
    const double dsq[3] = {ds2, ds3, ds4};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const double mp = 0.5 * dsq[q] * invN;
      out16[4 * q] = (float)mp;
      if (q == 1) mp1 = mp;
    }
    const double tr = (double)tr_coeff * (mp1 + (include_cov ? (double)o5 : 0.0));
    out16[13] = (float)tr;
    const float sur = (float)(-ds0 * invN);
    const float entl = ent_coef == 0.f ? 0.f : -ent_coef * o12;
    rec[0] = sur;
    rec[1] = entl;
    rec[2] = (float)tr;
    rec[3] = ent_coef == 0.f ? sur + (float)tr : sur + (float)tr + entl;
    rec[4] = o12;
    rec[5] = before;
    rec[6] = before * cf;
    if (rec_kl > 0) {
      // kl_old_new_proj of this epoch: the mean parts from the row sums (the
      // cov / shape / volume parts were put into this row by the kernel that
      // built the epoch's factors: bb_diag_fwd_kernel / the previous finish)
#pragma unroll
      for (int q = 0; q < 3; ++q) rec[7 + 4 * q] = (float)(0.5 * dsq[q] * invN);
    }
  }
  // ---- Adam on the registers
  float step_size, bc2s;
  adam_coef(lr, b1, b2, step, step_size, bc2s);
#pragma unroll
  for (int q = 0; q < NPT; ++q) {
    const int p = tid + q * SBT;
    float wn = pw[q], mi = pm[q], vi = pv[q];
    adam_elem(gr[q] * coef, wn, mi, vi, b1, b2, eps, wd, step_size, bc2s);
    if (p < P) {
      param[p] = wn;
      m[p] = mi;
      v[p] = vi;
    }
  }
  if (!w0) return;
  float var_n = var;
  if (live) {
    adam_elem(gvar * coef, var_n, mv, vv, b1, b2, eps, wd, step_size, bc2s);
    param[P + tid] = var_n;
    m[P + tid] = mv;
    v[P + tid] = vv;
  }
  if (!next_fwd) return;
  // ---- the next epoch's Cholesky head and covariance projection (wave 0)
  double sig = 1, so = 1;
  if (live) {
    sig = (double)((var_n > 20.f ? var_n : log1pf(expf(var_n))) + min_std);
    so = (double)so_f;
  }
  const double a = sig / so, lm = a * a;
  const double kl0 = 0.5 * wave_sum_f64(live ? lm - 1.0 - log(lm) : 0.0);
  const bool act_n = kl0 > eps_cov;
  double eta_n = 0, mu_n = lm;
  if (act_n) {
    double lo = 0.0, hi = -1.0;                          // hi < 0: no upper end seen yet
#pragma unroll 1
    for (int it = 0; it < 60; ++it) {
      double t = 0, dt = 0;
      if (live) {
        const double w = 1.0 / (eta_n * lm + 1.0);
        const double m_ = (eta_n + 1.0) * lm * w;
        t = m_ - 1.0 - log(m_);
        dt = (1.0 - 1.0 / m_) * lm * (1.0 - lm) * w * w;
      }
      const double h = 0.5 * wave_sum_f64(t), dh = 0.5 * wave_sum_f64(dt);
      if (h > eps_cov) lo = eta_n; else hi = eta_n;
      if (fabs(h - eps_cov) <= 1e-11 * eps_cov) break;
      double nxt = eta_n + 2.0 * h * (1.0 - sqrt(h / eps_cov)) / dh;
      if (!(nxt > lo && (hi < 0.0 || nxt < hi))) nxt = hi < 0.0 ? 2.0 * lo + 1.0 : 0.5 * (lo + hi);
      const bool done = fabs(nxt - eta_n) <= 1e-11 * fabs(nxt);
      eta_n = nxt;
      if (done) break;
    }
    mu_n = (eta_n + 1.0) * lm / (eta_n * lm + 1.0);
  }
  const double pl = act_n ? so * sqrt(mu_n) : sig;
  if (live) {
    L_new_w[kk] = (float)sig;
    L_proj_w[kk] = (float)pl;
    Li_proj_w[kk] = 1.f / (float)pl;
    gL_tr_w[kk] = include_cov ? (float)((double)tr_coeff * (sig / (pl * pl) - 1.0 / sig)) : 0.f;
    dctx_w[tid] = lm;
    dctx_w[K + tid] = mu_n;
  }
  const double r0 = sig / so, r1 = sig / pl, r2 = pl / so;
  const double f0 = wave_sum_f64(live ? r0 * r0 : 0.0), l0 = wave_sum_f64(live ? log(r0) : 0.0);
  const double f1 = wave_sum_f64(live ? r1 * r1 : 0.0), l1 = wave_sum_f64(live ? log(r1) : 0.0);
  const double f2 = wave_sum_f64(live ? r2 * r2 : 0.0), l2 = wave_sum_f64(live ? log(r2) : 0.0);
  const double lpld = wave_sum_f64(live ? log(pl) : 0.0);
  if (tid == 0) {
    const double f[3] = {f0, f1, f2}, l[3] = {l0, l1, l2};
    for (int q = 0; q < 3; ++q) {
      const double shape = 0.5 * (f[q] - (double)K), volume = -l[q];
      out16[4 * q + 0] = 0.f;
      out16[4 * q + 1] = (float)(shape + volume);
      out16[4 * q + 2] = (float)shape;
      out16[4 * q + 3] = (float)volume;
      if (rec_kl > 0) {                                  // the NEXT epoch's record row
        float* nx = rec + rec_kl + 7;
        nx[4 * q + 1] = (float)(shape + volume);
        nx[4 * q + 2] = (float)shape;
        nx[4 * q + 3] = (float)volume;
      }
    }
    out16[12] = (float)(0.5 * (double)K * (1.0 + 1.8378770664093453) + lpld);
    out16[13] = 0.f;
    out16[14] = 0.f;
    out16[15] = 0.f;
    dctx_w[2 * K] = eta_n;
    dctx_w[2 * K + 1] = act_n ? 1.0 : 0.0;
  }
}

// ---------------------------------------------------------------------------
// Skinny linear layer on rows: y [N][dout] = x [N][din] A (+ b), A = W^T for a
// torch Linear weight W [dout][din] (forward of an output layer) or A = W for
// W [din][dout] (the input gradient dX = dY W of the same layer) -- the two
// products around the policy mean net's output layer [N,128] <-> [N,K] that
// were the last library GEMMs of a TCE policy epoch.  Lane = row, the four
// waves split the outputs (OPW each, in registers), A sits in LDS as
// [din][4 OPW] and is read as broadcast 16-byte pieces, the x tile as
// row-major [64][pitch 4 x odd].
template <int OPW>
__global__ __launch_bounds__(SBT) void lin_rows_kernel(const float* __restrict__ x,
                                                       int64_t x_stride, int64_t N, int din,
                                                       int dout, const float* __restrict__ W,
                                                       int transposed,
                                                       const float* __restrict__ bias,
                                                       float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float S[];
  constexpr int DP = 4 * OPW;                              // padded outputs
  const int dinp = s_up4(din), xp = s_pitch(din);
  float* As = S;                                           // [dinp][DP]
  float* Bs = As + dinp * DP;                              // [DP]
  float* Xs = Bs + DP;                                     // [64][xp]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // A and the first x tile by LDS-DMA (s_dma64: one round trip instead of two
  // dependent load / store loops).  Pad
  // outputs (o >= dout) hold copies of the last output's weights -- their
  // accumulators are never stored --, pad inputs (rows din .. dinp - 1 of A)
  // are zeroed behind the wait, pad columns / rows past N of x hold copies.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned ulane = (unsigned)lane, udin = (unsigned)din, udout = (unsigned)dout;
  for (int t = wave_u; t < dinp * DP / 64; t += SNW) {       // (dinp DP: a multiple of 64)
    const unsigned a = 64u * t + ulane, i = a / DP, o = a % DP;
    const unsigned ic = i < udin ? i : udin - 1u, oc = o < udout ? o : udout - 1u;
    s_dma64(W + (transposed ? oc * udin + ic : ic * udout + oc), As + 64 * t);
  }
  auto x_tile = [&](int64_t r0) {
    const unsigned last = (unsigned)(N - r0 < SR ? N - r0 : SR) - 1u;
    const float* __restrict__ xb = x + r0 * x_stride;
    const unsigned stride = (unsigned)x_stride, uxp = (unsigned)xp, cl = udin - 1u;
    // (xp up to 260: a / xp by a 32-bit reciprocal, exact while a xp < 2^32)
    const unsigned mx = (unsigned)(((1ull << 32) + uxp - 1u) / uxp);
    for (int t = wave_u; t < xp; t += SNW) {
      const unsigned a = 64u * t + ulane, r = __umulhi(a, mx), c = a - r * uxp;
      s_dma64(xb + (r < last ? r : last) * stride + (c < cl ? c : cl), Xs + 64 * t);
    }
  };
  x_tile((int64_t)blockIdx.x * SR);
  for (int e = tid; e < DP; e += SBT) Bs[e] = (bias && e < dout) ? bias[e] : 0.f;
  s_dma_wait();
  for (int e = din * DP + tid; e < dinp * DP; e += SBT) As[e] = 0.f;
  bool first = true;
  for (int64_t r0 = (int64_t)blockIdx.x * SR; r0 < N; r0 += (int64_t)gridDim.x * SR) {
    if (!first) {
      __syncthreads();
      x_tile(r0);
      s_dma_wait();
    }
    first = false;
    __syncthreads();
    float acc[OPW];
#pragma unroll
    for (int u = 0; u < OPW; ++u) acc[u] = Bs[wave * OPW + u];
    for (int i0 = 0; i0 < dinp; i0 += 4) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(Xs + lane * xp + i0);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float* ar = As + (i0 + t) * DP + wave * OPW;
#pragma unroll
        for (int u = 0; u < OPW; u += 4) {
          const f32x4 av = *reinterpret_cast<const f32x4*>(ar + u);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[u + q] += av[q] * xv[t];
        }
      }
    }
    // through LDS for coalesced row stores: y tile [64][dout]
    __syncthreads();
    float* Ys = Xs;                                        // reuse (xp >= ... see host check)
    const int yp = s_pitch(DP);
#pragma unroll
    for (int u = 0; u < OPW; u += 4)
      *reinterpret_cast<f32x4*>(Ys + lane * yp + wave * OPW + u) =
          f32x4{acc[u], acc[u + 1], acc[u + 2], acc[u + 3]};
    __syncthreads();
    for (int e = tid; e < SR * dout; e += SBT) {
      const int r = e / dout, o = e - r * dout;
      if (r0 + r < N) y[(r0 + r) * dout + o] = Ys[r * yp + o];
    }
  }
}

inline int s_grid(int64_t N) { return (int)tmin<int64_t>(ceil_div(N, SR), S_MAX_GRID); }

template <int OPW>
int lin_rows_launch(const float* x, int64_t x_stride, int64_t N, int din, int dout, const float* W,
                    int transposed, const float* bias, float* y, hipStream_t st) {
  constexpr int DP = 4 * OPW;
  const int tile = SR * (s_pitch(din) > s_pitch(DP) ? s_pitch(din) : s_pitch(DP));
  const size_t lds = sizeof(float) * ((size_t)s_up4(din) * DP + DP + tile);
  TCE_CHECK_ARG(lds <= S_LDS_MAX, "lin_rows: shape does not fit the LDS");
  tce_lds_limit(reinterpret_cast<const void*>(lin_rows_kernel<OPW>), lds);
  hipLaunchKernelGGL(lin_rows_kernel<OPW>, dim3(s_grid(N)), dim3(SBT), lds, st, x, x_stride, N,
                     din, dout, W, transposed, bias, y);
  TCE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------
inline bool s_shape_ok(int din, int H, int dout) {
  return din >= 1 && din <= 64 && (H == 32 || H == 64) && dout >= 1 && dout <= 64;
}
inline size_t s_lds_bytes(int din, int H, int dout, int head) {
  if (head == HEAD_NONE) return sizeof(float) * (size_t)s_forward_lds_floats(din, H, dout);
  return sizeof(float) * (size_t)s_lds(din, H, dout, head).total;
}

template <int H, int ACT, int HEAD, int KP, bool DG = false>
int s_launch_epoch(const SNet& n, const SValueHead& vh, const SPolicyHead& ph, const SReduce& rd,
                   hipStream_t st) {
  const size_t lds = s_lds_bytes(n.din, H, n.dout, HEAD);
  tce_lds_limit(reinterpret_cast<const void*>(smlp_epoch_kernel<H, ACT, HEAD, KP, DG>), lds);
  hipLaunchKernelGGL((smlp_epoch_kernel<H, ACT, HEAD, KP, DG>), dim3(s_grid(n.N)), dim3(SBT), lds,
                     st, n, vh, ph, rd);
  TCE_LAUNCH_CHECK();
  return 0;
}
// value head: one variant per (H, activation); policy head: one per (H, padded
// K) with the activation as a kernel argument
int s_dispatch_value(int H, int act, const SNet& n, const SValueHead& vh, const SReduce& rd,
                     hipStream_t st) {
  const SPolicyHead ph{};
#define S_CASE(HH, AA) \
  if (H == HH && act == AA) return s_launch_epoch<HH, AA, HEAD_VALUE, 0>(n, vh, ph, rd, st);
  S_CASE(32, ACT_TANH) S_CASE(32, ACT_RELU) S_CASE(32, ACT_LEAKY) S_CASE(32, ACT_SOFTPLUS)
  S_CASE(64, ACT_TANH) S_CASE(64, ACT_RELU) S_CASE(64, ACT_LEAKY) S_CASE(64, ACT_SOFTPLUS)
#undef S_CASE
  tce_set_error("smlp: unsupported hidden width / activation");
  return 1;
}
int s_dispatch_policy(int H, const SNet& n, const SPolicyHead& ph, const SReduce& rd, bool diag,
                      hipStream_t st) {
  const SValueHead vh{};
  const int K = n.dout;
  const int kp = K <= 8 ? 8 : K <= 16 ? 16 : K <= 24 ? 24 : K <= 32 ? 32 : 0;
  // diagonal factors (K <= 32): the head without the K x K images
#define S_CASE(HH, KK) \
  if (diag && H == HH && kp == KK) \
    return s_launch_epoch<HH, ACT_RT, HEAD_BB_POLICY, KK, true>(n, vh, ph, rd, st);
  S_CASE(32, 8) S_CASE(32, 16) S_CASE(32, 24) S_CASE(32, 32)
  S_CASE(64, 8) S_CASE(64, 16) S_CASE(64, 24) S_CASE(64, 32)
#undef S_CASE
#define S_CASE(HH, KK) \
  if (H == HH && kp == KK) return s_launch_epoch<HH, ACT_RT, HEAD_BB_POLICY, KK>(n, vh, ph, rd, st);
  S_CASE(32, 8) S_CASE(32, 16) S_CASE(32, 24) S_CASE(32, 32) S_CASE(32, 0)
  S_CASE(64, 8) S_CASE(64, 16) S_CASE(64, 24) S_CASE(64, 32) S_CASE(64, 0)
#undef S_CASE
  tce_set_error("smlp: unsupported hidden width");
  return 1;
}

template <int H, int ACT>
int s_launch_forward(const SNet& n, float* out, hipStream_t st) {
  const size_t lds = s_lds_bytes(n.din, H, n.dout, HEAD_NONE);
  tce_lds_limit(reinterpret_cast<const void*>(smlp_forward_kernel<H, ACT>), lds);
  hipLaunchKernelGGL((smlp_forward_kernel<H, ACT>), dim3(s_grid(n.N)), dim3(SBT), lds, st, n, out);
  TCE_LAUNCH_CHECK();
  return 0;
}

inline void s_reduce_ws(float* ws, int64_t N, int din, int H, int dout, SReduce& rd) {
  const int64_t PS = s_up4(s_nparams(din, H, dout) + dout * dout);
  const int64_t g = s_grid(N);
  rd.slabs = ws;
  rd.dpart = reinterpret_cast<double*>(ws + g * PS);
  rd.PS = (int)PS;
  rd.P = s_nparams(din, H, dout);
}

inline int s_launch_reduce(const SFinish& f, hipStream_t st) {
  const unsigned grid = (unsigned)ceil_div(f.P + f.KK, 64);
  hipLaunchKernelGGL(smlp_reduce_kernel, dim3(grid), dim3(64 * FIN_GROUPS), 0, st, f);
  TCE_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" {

int tce_smlp_supported(int din, int H, int dout, int head) {
  if (!s_shape_ok(din, H, dout) || head < 0 || head > 2) return 0;
  return s_lds_bytes(din, H, dout, head) <= S_LDS_MAX ? 1 : 0;
}

int64_t tce_smlp_num_params(int din, int H, int dout) { return s_nparams(din, H, dout); }

// floats: gradient slabs [grid][P + dout^2 (+ pad)], then doubles [grid][8] and 8 more
int64_t tce_smlp_ws_len(int64_t N, int din, int H, int dout) {
  const int64_t PS = s_up4(s_nparams(din, H, dout) + dout * dout);
  const int64_t g = s_grid(N);
  return g * PS + 2 * (g * 8) + 2 * 8 + 8 + 8;   // ... + dsum + zero3 + 8 spare (stamps)
}

int tce_smlp_forward_f32(const float* x, int64_t x_stride, int64_t N, int din, int H, int dout,
                         int act, const float* param, float* out, void* stream) {
  TCE_CHECK_ARG(x && param && out && N > 0, "smlp_forward: null buffer / empty batch");
  TCE_CHECK_ARG(tce_smlp_supported(din, H, dout, HEAD_NONE), "smlp_forward: unsupported shape");
  TCE_CHECK_ARG(x_stride >= din, "smlp_forward: row stride < din");
  const SNet n{x, x_stride, N, din, dout, param, act};
  hipStream_t st = (hipStream_t)stream;
#define S_CASE(HH, AA) \
  if (H == HH && act == AA) return s_launch_forward<HH, AA>(n, out, st);
  S_CASE(32, ACT_TANH) S_CASE(32, ACT_RELU) S_CASE(32, ACT_LEAKY) S_CASE(32, ACT_SOFTPLUS)
  S_CASE(64, ACT_TANH) S_CASE(64, ACT_RELU) S_CASE(64, ACT_LEAKY) S_CASE(64, ACT_SOFTPLUS)
#undef S_CASE
  tce_set_error("smlp_forward: unsupported hidden width / activation");
  return 1;
}

int tce_smlp_critic_epochs_f32(const float* x, int64_t x_stride, const float* returns,
                               const float* old_values, int64_t N, int din, int H, int act,
                               float clip_critic, float* param, float* grad, float* m, float* v,
                               float* opt_state, float lr, float beta1, float beta2, float eps,
                               float weight_decay, float clip_grad, float grad_scale, int do_adam,
                               int first_step, int epochs, float* ws, float* rec, void* xchg,
                               void* stream) {
  TCE_CHECK_ARG(x && returns && param && grad && ws && rec && N > 0 && epochs > 0,
                "smlp_critic_epochs: null buffer / empty batch");
  TCE_CHECK_ARG(!xchg || do_adam, "smlp_critic_epochs: an exchange needs the Adam step");
  TCE_CHECK_ARG(!(clip_critic > 0.f) || old_values, "smlp_critic_epochs: old values missing");
  TCE_CHECK_ARG(!do_adam || (m && v && opt_state), "smlp_critic_epochs: optimizer state missing");
  TCE_CHECK_ARG(tce_smlp_supported(din, H, 1, HEAD_VALUE), "smlp_critic_epochs: unsupported shape");
  TCE_CHECK_ARG(x_stride >= din && x_stride < (1 << 24), "smlp_critic_epochs: row stride outside [din, 2^24)");
  TCE_CHECK_ARG(do_adam || epochs == 1, "smlp_critic_epochs: epochs > 1 needs the Adam step");
  hipStream_t st = (hipStream_t)stream;
  const SNet n{x, x_stride, N, din, 1, param, act};
  const SValueHead vh{returns, old_values, clip_critic};
  SReduce rd{};
  s_reduce_ws(ws, N, din, H, 1, rd);
  const int g = s_grid(N);
  double* dsum = rd.dpart + (int64_t)g * 8;
  // the Adam step rides on the slab reduction unless the clip factor needs the norm
  // first -- or the envs are sharded: the reduction grid is tens of workgroups, and
  // workgroups that wait for a peer must be few (csrc/mlp_shared.h); the exchange +
  // Adam follow as one small launch
  const bool fuse = do_adam && !(clip_grad > 0.f) && !xchg;
  for (int e = 0; e < epochs; ++e) {
    int rc = s_dispatch_value(H, act, n, vh, rd, st);
    if (rc) return rc;
    SFinish f{};
    f.slabs = rd.slabs; f.dpart = rd.dpart; f.nparts = g; f.PS = rd.PS; f.P = rd.P; f.KK = 0;
    f.N = N; f.grad = grad; f.extra = nullptr; f.dsum = dsum; f.stats = rec + 3 * e;
    if (fuse) {
      f.param = param; f.m = m; f.v = v; f.state = opt_state;
      f.lr = lr; f.b1 = beta1; f.b2 = beta2; f.eps = eps; f.wd = weight_decay;
      f.step = (float)(first_step + e); f.gscale = grad_scale;
    }
    rc = s_launch_reduce(f, st);
    if (rc) return rc;
    if (do_adam && xchg) {
      // rec row: {mean loss (local), |g| of the rank-averaged gradient, the same clipped}
      rc = tce_xchg_adam_f32(xchg, param, grad, m, v, rd.P, opt_state, rec + 3 * e + 1,
                             (float)(first_step + e), lr, beta1, beta2, eps, weight_decay,
                             clip_grad, grad_scale, stream);
      if (rc) return rc;
    } else if (do_adam && !fuse) {
      rc = tce_adam_flat_f32(param, grad, m, v, rd.P, opt_state, rec + 3 * e + 1, lr, beta1, beta2,
                             eps, weight_decay, clip_grad, grad_scale, stream);
      if (rc) return rc;
    }
  }
  return 0;
}

// (tests: the general finish kernel for diagonal factors too)
static int g_bb_finish_general = 0;
void tce_bb_finish_general(int on) { g_bb_finish_general = on; }

int tce_bb_policy_epochs_f32(const float* x, int64_t x_stride, const float* actions,
                             const float* logp_old, const float* adv, const float* mean_old,
                             const float* L_old, int64_t N, int din, int H, int K, int act,
                             int nvec, float min_std, float eps_mean, double eps_cov,
                             const float* beta, int entropy_eq, float tr_coeff,
                             int tr_include_cov, float ent_coef, float* param, float* grad,
                             float* m, float* v, float* opt_state, float lr, float beta1,
                             float beta2, float eps, float weight_decay, float clip_grad,
                             float grad_scale, int do_adam, int diag, int epochs,
                             double* proj_ctx, float* ws, float* mats, float* rec,
                             int rec_stride, float* mean_new_out, float* proj_mean_out,
                             void* xchg, void* stream) {
  TCE_CHECK_ARG(!xchg || do_adam, "bb_policy_epochs: an exchange needs the Adam step");
  TCE_CHECK_ARG(rec_stride == 7 || rec_stride >= 19,
                "bb_policy_epochs: rec_stride is 7 (losses / norms) or >= 19 (+ 12 KL means)");
  const int rec_kl = rec_stride >= 19;
  TCE_CHECK_ARG(x && actions && logp_old && adv && mean_old && L_old && param && grad &&
                    proj_ctx && ws && mats && rec && N > 0 && epochs > 0,
                "bb_policy_epochs: null buffer / empty batch");
  TCE_CHECK_ARG(!do_adam || (m && v && opt_state), "bb_policy_epochs: optimizer state missing");
  TCE_CHECK_ARG(tce_smlp_supported(din, H, K, HEAD_BB_POLICY), "bb_policy_epochs: unsupported shape");
  TCE_CHECK_ARG(nvec == K || nvec == K + K * (K - 1) / 2, "bb_policy_epochs: bad variance vector");
  TCE_CHECK_ARG(!(diag & 1) || (nvec == K && beta == nullptr),
                "bb_policy_epochs: the diagonal path needs std_only and no entropy bound");
  TCE_CHECK_ARG(x_stride >= din && x_stride < (1 << 24), "bb_policy_epochs: row stride outside [din, 2^24)");
  TCE_CHECK_ARG(do_adam || epochs == 1, "bb_policy_epochs: epochs > 1 needs the Adam step");
  hipStream_t st = (hipStream_t)stream;
  const int P = s_nparams(din, H, K);
  const int KK = s_up4(K * K);
  // mats: L_new | L_proj | g_pL | gL_tr | gL_p | Li_old | Li_proj  [K,K] each, then out16 [16]
  float* L_new = mats;
  float* L_proj = L_new + KK;
  float* g_pL = L_proj + KK;
  float* gL_tr = g_pL + KK;
  float* gL_p = gL_tr + KK;
  float* Li_old = gL_p + KK;
  float* Li_proj = Li_old + KK;
  float* out16 = Li_proj + KK;
  // (diag bit 1: `mats` still holds L_old^-1 from an earlier call of this update --
  // the per-epoch calls of a balance-check iteration)
  const bool have_Li_old = (diag & 2) != 0;
  diag &= 1;
  if (!have_Li_old) {
    const int rc0 = s_tri_inverse(L_old, Li_old, K, st);     // once per update
    if (rc0) return rc0;
  }
  TCE_CHECK_ARG(act >= 0 && act <= 3, "bb_policy_epochs: unknown activation");
  SNet n{x, x_stride, N, din, K, param, act};
  SReduce rd{};
  s_reduce_ws(ws, N, din, H, K, rd);
  const int g = s_grid(N);
  double* dsum = rd.dpart + (int64_t)g * 8;
  double* zero3 = dsum + 8;                       // stays zero: the K x K parts without the mean sums
  for (int e = 0; e < epochs; ++e) {
    const bool lastep = e == epochs - 1;
    int rc;
    // (diagonal factors: from the second epoch on the previous finish kernel has
    // already run this epoch's head / projection behind its Adam step)
    const bool chained = diag && do_adam;
    if (diag) {
      if (e == 0 || !chained) {
        hipLaunchKernelGGL(bb_diag_fwd_kernel, dim3(1), dim3(64), 0, st, param + P, min_std, L_old,
                           K, eps_cov, tr_coeff, tr_include_cov, L_new, L_proj, Li_proj, gL_tr,
                           out16, proj_ctx,
                           rec_kl ? rec + (int64_t)rec_stride * e + 7 : (float*)nullptr);
        TCE_LAUNCH_CHECK();
      }
    } else {
      rc = tce_chol_build_fwd_f32(param + P, L_new, 1, K, nvec, min_std, stream);
      if (rc) return rc;
      rc = tce_kl_cov_proj_fwd_f32(L_new, L_old, 0, eps_cov, beta, entropy_eq, L_proj, proj_ctx,
                                   1, K, 1, stream);
      if (rc) return rc;
      rc = tce_kl_shared_mat_f32(L_new, L_old, L_proj, N, K, tr_coeff, tr_include_cov, zero3, 1,
                                 out16, gL_tr, stream);
      if (rc) return rc;
      if (K <= 32) {                                 // the register head works with the inverse
        rc = s_tri_inverse(L_proj, Li_proj, K, st);
        if (rc) return rc;
      }
    }
    const SPolicyHead ph{actions, logp_old, adv, mean_old, L_old, L_proj, Li_old, Li_proj,
                         eps_mean, tr_coeff,
                         ent_coef, lastep ? mean_new_out : nullptr,
                         lastep ? proj_mean_out : nullptr};
    rc = s_dispatch_policy(H, n, ph, rd, diag != 0, st);
    if (rc) return rc;
    SFinish f{};
    f.slabs = rd.slabs; f.dpart = rd.dpart; f.nparts = g; f.PS = rd.PS; f.P = P; f.KK = K * K;
    f.N = N; f.grad = grad; f.extra = g_pL; f.dsum = dsum; f.stats = nullptr;
    rc = s_launch_reduce(f, st);
    if (rc) return rc;
    if (!diag) {
      hipLaunchKernelGGL(bb_policy_finish_kernel, dim3(1), dim3(SBT), 0, st, 0, 0, g_pL, gL_tr,
                         gL_p, L_proj, L_old, proj_ctx, dsum, N, K, nvec, P, tr_coeff,
                         tr_include_cov, ent_coef, param, grad, m, v, opt_state, lr, beta1, beta2,
                         eps, weight_decay, clip_grad, grad_scale, do_adam, out16,
                         rec + (int64_t)rec_stride * e, 0, 0, 0.f, 0.0, nullptr, nullptr, nullptr,
                         nullptr, nullptr, xchg_none());
      TCE_LAUNCH_CHECK();
      rc = tce_kl_cov_proj_bwd_f32(L_new, L_old, 0, L_proj, proj_ctx, g_pL, gL_p, 1, K, stream);
      if (rc) return rc;
    }
    XchgView X;                                      // the epoch's gradient exchange (env shards)
    if (xchg_next(do_adam ? xchg : nullptr, (int64_t)(P + nvec) * 4, 1, &X)) return 1;
    if (diag && do_adam && P <= SBT * FD_NPT && !g_bb_finish_general) {
      const int npt = (P + SBT - 1) / SBT;
#define FD_LAUNCH(NN)                                                                             \
  hipLaunchKernelGGL(bb_diag_finish_kernel<NN>, dim3(1), dim3(SBT), 0, st, g_pL, gL_tr, L_proj,   \
                     L_old, proj_ctx, dsum, N, K, P, tr_coeff, tr_include_cov, ent_coef, param,   \
                     grad, m, v, opt_state, lr, beta1, beta2, eps, weight_decay, clip_grad,       \
                     grad_scale, out16, rec + (int64_t)rec_stride * e, rec_kl ? rec_stride : 0,   \
                     chained && !lastep ? 1 : 0, min_std, eps_cov, L_new, L_proj, Li_proj, gL_tr, \
                     proj_ctx, X)
      if (npt <= 4) FD_LAUNCH(4);
      else if (npt <= 8) FD_LAUNCH(8);
      else if (npt <= 12) FD_LAUNCH(12);
      else FD_LAUNCH(16);
#undef FD_LAUNCH
      TCE_LAUNCH_CHECK();
      continue;
    }
    hipLaunchKernelGGL(bb_policy_finish_kernel, dim3(1), dim3(SBT), 0, st, 1, diag, g_pL, gL_tr,
                       gL_p, L_proj, L_old, proj_ctx, dsum, N, K, nvec, P, tr_coeff, tr_include_cov,
                       ent_coef, param, grad, m, v, opt_state, lr, beta1, beta2, eps, weight_decay,
                       clip_grad, grad_scale, do_adam, out16, rec + (int64_t)rec_stride * e, rec_kl,
                       chained && !lastep ? 1 : 0, min_std, eps_cov, L_new, L_proj, Li_proj, gL_tr,
                       proj_ctx, X);
    TCE_LAUNCH_CHECK();
  }
  return 0;
}

int tce_lin_rows_f32(const float* x, int64_t x_stride, int64_t N, int din, int dout,
                     const float* W, int transposed, const float* bias, float* y, void* stream) {
  TCE_CHECK_ARG(x && W && y && N > 0 && din >= 1 && din <= 256 && dout >= 1 && dout <= 128 &&
                    x_stride >= din && x_stride < (1 << 24),
                "lin_rows: bad arguments (D_in <= 256, D_out <= 128, row stride in [D_in, 2^24))");
  hipStream_t st = (hipStream_t)stream;
  if (dout <= 16) return lin_rows_launch<4>(x, x_stride, N, din, dout, W, transposed, bias, y, st);
  if (dout <= 32) return lin_rows_launch<8>(x, x_stride, N, din, dout, W, transposed, bias, y, st);
  if (dout <= 64) return lin_rows_launch<16>(x, x_stride, N, din, dout, W, transposed, bias, y, st);
  return lin_rows_launch<32>(x, x_stride, N, din, dout, W, transposed, bias, y, st);
}

int64_t tce_bb_policy_mats_len(int K) { return 7 * (int64_t)s_up4(K * K) + 16; }

}  // extern "C"
