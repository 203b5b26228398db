// Fused critic MLP epoch for gfx950: forward + MSE value loss + backward +
// per-workgroup gradient partials in ONE pass over the rollout states, on
// exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//
// Replaces, per critic epoch (mprl/rl/agent/temporal_correlated_agent.py:343-366):
//   values_new = critic(states[..., :-2 dof])         mprl/util/util_nn.py:225-246
//   loss = value_loss(values_new, returns, old_vs)    :688-716
//   loss.backward()
// for the 2-hidden-layer value network  D_in -> 128 -> 128 -> 1  of the
// Metaworld config (hidden width 128; other widths use the library-GEMM path).
//
// Orientation: everything is computed TRANSPOSED, activations as [hidden x
// batch].  An MFMA result tile (batch column on the lane, 4 hidden rows in the
// registers) is then directly the B operand of the next layer's MFMA (which
// contracts over the hidden index), so the forward chain X -> H1 -> H2 -> v and
// the backward chain dY2 -> dH1 never leave the registers; only the weights (A
// operands) come from LDS.  The weight gradients contract over the batch index,
// which sits on the lanes: for those the tiles are written once to LDS as
// [batch][hidden] and re-read as A/B fragments; the 4 waves of a workgroup
// split the output rows of dW2 / dW1.
//
// LDS traffic is what bounds a 1-wave-per-SIMD MFMA kernel, so every operand
// stream is a conflict-free 16-byte read feeding 4..8 MFMAs, fetched one step
// ahead of its use (explicit double buffers + sched_barrier):
//  * hidden layer 1 lives in a permuted order.  Block kb, row r of an MFMA tile
//    holds unit  u1 = 64 (kb>>2) + 16 (r&3) + 4 (r>>2) + (kb&3).  With W2 kept
//    row-major [h2][h1] at pitch 136, the forward read (row h2, 4 consecutive
//    kb at fixed r) and the backward read (row h2 = k, 4 consecutive kb at
//    fixed output row r) are both contiguous float4 and hit 16 distinct
//    16-byte bank slots per 16-lane group.
//  * the [batch][hidden] transposes use pitch 132: float4 stores from the C/D
//    layout, float4 / float2 loads for dW2 / dW1 with the output blocks
//    interleaved (column 8 c + n, row 32 wave + 2 c + rb), all conflict free.
//  * db1 comes out of the dW1 MFMAs through a column of ones appended to X.
//
// Work decomposition: a workgroup = 4 waves = 64 batch rows per tile (16 per
// wave), persistent over its share of the tiles; weight-gradient accumulators
// stay in registers across tiles; one partial slab per workgroup at the end,
// reduced by mlp_reduce_kernel.  MFMA-bound: 944 MFMAs per wave and tile.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int HID = 128;
constexpr int NB = HID / 16;          // 8 blocks of 16 hidden units
constexpr int W2P = 136;              // LDS pitch of W2 (see header)
constexpr int TPT = 132;              // pitch of the [64 batch][hidden] transposes
constexpr int XP = 68;                // pitch of the X tile stashed for dW1
constexpr int MLP_BT = 256;
constexpr int ROWS_PER_TILE = 64;
constexpr int MAX_DIN = 40;

enum { ACT_TANH = 0, ACT_RELU = 1, ACT_LEAKY = 2, ACT_SOFTPLUS = 3 };

template <int ACT>
__device__ inline float act_f(float y) {
  if (ACT == ACT_TANH) return tanhf(y);
  if (ACT == ACT_RELU) return y > 0.f ? y : 0.f;
  if (ACT == ACT_LEAKY) return y > 0.f ? y : 0.01f * y;
  return y > 20.f ? y : log1pf(expf(y));
}
// derivative expressed with the OUTPUT h = act(y)
template <int ACT>
__device__ inline float act_d(float h) {
  if (ACT == ACT_TANH) return 1.f - h * h;
  if (ACT == ACT_RELU) return h > 0.f ? 1.f : 0.f;
  if (ACT == ACT_LEAKY) return h > 0.f ? 1.f : 0.01f;
  return 1.f - expf(-h);
}

__device__ inline f32x4 mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// nothing moves across: keeps the next step's LDS reads ahead of this step's MFMAs
__device__ inline void fence_sched() { __builtin_amdgcn_sched_barrier(0); }

// hidden-1 unit held at position q = 16 kb + r of the MFMA tiles
__host__ __device__ inline int u1_of(int q) {
  const int kb = q >> 4, r = q & 15;
  return 64 * (kb >> 2) + 16 * (r & 3) + 4 * (r >> 2) + (kb & 3);
}

struct MlpArgs {
  const float* x;        // states, row r = (n, t): x + (n * env_stride + t * row_stride)
  int64_t env_stride, row_stride;
  int T;                 // rows per env
  int64_t R;             // total rows
  int din;               // input features used (first din of each row)
  const float *w1, *b1, *w2, *b2, *w3, *b3;   // torch Linear layout [out][in]
  const float* ret;      // returns [R]
  const float* old_v;    // old values [R] (clipped loss) or nullptr
  float clip;            // clip_critic (<= 0: plain MSE)
  float* values;         // forward output [R] (nullable)
  float* partials;       // [gridDim.x][P + 2] gradient slabs (+ loss sum, count) (nullable: forward only)
};

__host__ __device__ inline int mlp_num_params(int din) {
  return HID * din + HID + HID * HID + HID + HID + 1;
}

// KPGE: input features per lane group (even); D_in <= 4 KPGE.
template <int ACT, bool BWD, int KPGE>
__global__ __launch_bounds__(MLP_BT, 1) void mlp_critic_kernel(MlpArgs a) {
  constexpr int W1P = 4 * KPGE + 2;                           // even pitch: float2 reads
  constexpr int NCB = (4 * KPGE + 1 + 15) / 16;               // dW1 column blocks incl. the ones column
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* W2s = reinterpret_cast<float*>(smem_raw);            // [HID][W2P]   rows h2, columns h1 unit
  float* W1s = W2s + HID * W2P;                               // [HID pos][W1P]
  float* Bs = W1s + HID * W1P;                                // b1 (by position) | b2 | w3
  float* Th1 = Bs + 3 * HID;                                  // [64][TPT] H1 by position; later X stash [64][XP]
  float* Tdy = Th1 + ROWS_PER_TILE * TPT;                     // [64][TPT] dY2 (by h2), later dY1 (by position)
  const int din = a.din;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;

  // ---- stage the weights once per workgroup
  for (int e = tid; e < HID * HID; e += MLP_BT) W2s[(e >> 7) * W2P + (e & 127)] = a.w2[e];
  for (int e = tid; e < HID * 4 * KPGE; e += MLP_BT) {
    const int q = e / (4 * KPGE), f = e - q * 4 * KPGE;
    W1s[q * W1P + f] = f < din ? a.w1[u1_of(q) * din + f] : 0.f;
  }
  for (int e = tid; e < HID; e += MLP_BT) {
    Bs[e] = a.b1[u1_of(e)];
    Bs[HID + e] = a.b2[e];
    Bs[2 * HID + e] = a.w3[e];
  }
  __syncthreads();
  const float b3 = a.b3[0];

  // persistent gradient accumulators
  f32x4 gW2[2][NB];        // rows q_a = 32 wave + 2 (4 g + i) + rb, columns q_b = 8 c + n
  f32x4 gW1[2][NCB];       // rows as gW2 (positions of hidden 1), columns feature 16 n + c
  float gb2[NB][4], gw3[NB][4];
  float gb3 = 0.f, loss_sum = 0.f;
  if (BWD) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
      for (int n = 0; n < NB; ++n) gW2[r][n] = (f32x4){0, 0, 0, 0};
#pragma unroll
      for (int n = 0; n < NCB; ++n) gW1[r][n] = (f32x4){0, 0, 0, 0};
    }
#pragma unroll
    for (int m = 0; m < NB; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) { gb2[m][i] = 0.f; gw3[m][i] = 0.f; }
  }

  const int64_t ntiles = (a.R + ROWS_PER_TILE - 1) / ROWS_PER_TILE;
  const float inv_n = 1.f / (float)a.R;
  float* trow_h = Th1 + (wave * 16 + c) * TPT + 4 * g;       // this lane's C/D rows in the transposes
  float* trow_d = Tdy + (wave * 16 + c) * TPT + 4 * g;

  // X fragment of a tile: lane (c, g) holds X[r][KPGE g + s], s < KPGE.  Loaded
  // one tile ahead (clamped addresses, no branches around the loads) so that
  // the HBM latency hides behind the previous tile's MFMAs.
  auto load_x = [&](int64_t tile, float* dst) {
    const int64_t rr = tile * ROWS_PER_TILE + wave * 16 + c;
    const int64_t rcl = rr < a.R ? rr : a.R - 1;
    const int64_t ne = rcl / a.T;
    const float* xr = a.x + ne * a.env_stride + (rcl - ne * a.T) * a.row_stride;
#pragma unroll
    for (int s = 0; s < KPGE; ++s) {
      const int k = KPGE * g + s;
      dst[s] = xr[k < din ? k : din - 1];
    }
  };
  float xn[KPGE];
  if ((int64_t)blockIdx.x < ntiles) load_x(blockIdx.x, xn);

  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * ROWS_PER_TILE;
    const int64_t r = row0 + wave * 16 + c;                   // this lane's batch row
    const bool rok = r < a.R;
    const int64_t rc = rok ? r : a.R - 1;

    // ---- F1: take the prefetched fragment, start the next tile's loads
    float xb[KPGE];
#pragma unroll
    for (int s = 0; s < KPGE; ++s) xb[s] = (KPGE * g + s < din && rok) ? xn[s] : 0.f;
    {
      const int64_t nt = tile + gridDim.x;
      load_x(nt < ntiles ? nt : tile, xn);
    }
    // ---- F2: Y1^T = W1 X^T + b1  (A = W1 rows by position, B = X fragment).
    // Two row blocks at a time: a dependent accumulate needs 40 cycles but the
    // MFMA issues every 32, so every chain is paired with a second one.
    f32x4 h1[NB];
    {
      float A[2][2 * KPGE];
      auto ld = [&](int mp, float* d) {
        const float* p = W1s + (32 * mp + c) * W1P + KPGE * g;
#pragma unroll
        for (int s = 0; s < KPGE; s += 2) {
          const f32x2 v0 = *reinterpret_cast<const f32x2*>(p + s);
          const f32x2 v1 = *reinterpret_cast<const f32x2*>(p + 16 * W1P + s);
          d[s] = v0.x; d[s + 1] = v0.y;
          d[KPGE + s] = v1.x; d[KPGE + s + 1] = v1.y;
        }
      };
      ld(0, A[0]);
#pragma unroll
      for (int mp = 0; mp < NB / 2; ++mp) {
        if (mp + 1 < NB / 2) ld(mp + 1, A[(mp + 1) & 1]);
        f32x4 acc0 = *reinterpret_cast<const f32x4*>(Bs + 32 * mp + 4 * g);
        f32x4 acc1 = *reinterpret_cast<const f32x4*>(Bs + 32 * mp + 16 + 4 * g);
        fence_sched();
#pragma unroll
        for (int s = 0; s < KPGE; ++s) {
          acc0 = mfma(A[mp & 1][s], xb[s], acc0);
          acc1 = mfma(A[mp & 1][KPGE + s], xb[s], acc1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc0[i] = act_f<ACT>(acc0[i]); acc1[i] = act_f<ACT>(acc1[i]); }
        h1[2 * mp] = acc0;
        h1[2 * mp + 1] = acc1;
      }
    }
    // ---- F4: Y2^T = W2 H1^T + b2.  Step (mp, j): rows 32 mp + c and + 16, the
    // 8 k-steps kb with B = h1[kb][j] (unit 64 (kb>>2) + 16 j + 4 g + (kb&3)).
    f32x4 h2[NB];
    {
      f32x4 A[2][4];
      const float* wb = W2s + c * W2P + 4 * g;
      auto ld = [&](int st, f32x4* d) {
        const float* p = wb + (32 * (st >> 2)) * W2P + 16 * (st & 3);
        d[0] = *reinterpret_cast<const f32x4*>(p);
        d[1] = *reinterpret_cast<const f32x4*>(p + 64);
        d[2] = *reinterpret_cast<const f32x4*>(p + 16 * W2P);
        d[3] = *reinterpret_cast<const f32x4*>(p + 16 * W2P + 64);
      };
      ld(0, A[0]);
      f32x4 acc0, acc1;
#pragma unroll
      for (int st = 0; st < 16; ++st) {
        const int mp = st >> 2, j = st & 3;
        if (st + 1 < 16) ld(st + 1, A[(st + 1) & 1]);
        if (j == 0) {
          acc0 = *reinterpret_cast<const f32x4*>(Bs + HID + 32 * mp + 4 * g);
          acc1 = *reinterpret_cast<const f32x4*>(Bs + HID + 32 * mp + 16 + 4 * g);
        }
        fence_sched();
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
          acc0 = mfma(A[st & 1][kb >> 2][kb & 3], h1[kb][j], acc0);
          acc1 = mfma(A[st & 1][2 + (kb >> 2)][kb & 3], h1[kb][j], acc1);
        }
        if (j == 3) {
#pragma unroll
          for (int i = 0; i < 4; ++i) { acc0[i] = act_f<ACT>(acc0[i]); acc1[i] = act_f<ACT>(acc1[i]); }
          h2[2 * mp] = acc0;
          h2[2 * mp + 1] = acc1;
        }
      }
    }
    // ---- F5: v = w3 . H2 + b3 (sum over the 4 lane groups of a column)
    float v = 0.f;
#pragma unroll
    for (int m = 0; m < NB; ++m) {
      const f32x4 w3v = *reinterpret_cast<const f32x4*>(Bs + 2 * HID + 16 * m + 4 * g);
#pragma unroll
      for (int i = 0; i < 4; ++i) v += w3v[i] * h2[m][i];
    }
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    v += b3;
    if (a.values && rok && g == 0) a.values[r] = v;
    if (!BWD) continue;

    // ---- loss and dL/dv (mean over ALL rows R of the epoch)
    const float rt = a.ret[rc];
    float dv;
    {
      const float e = v - rt;
      float l = e * e, d = 2.f * e;
      if (a.clip > 0.f) {
        const float ov = a.old_v[rc];
        const float dlt = v - ov;
        const float cl = fminf(fmaxf(dlt, -a.clip), a.clip);
        const float e2 = ov + cl - rt;
        if (e2 * e2 > l) { l = e2 * e2; d = (dlt > -a.clip && dlt < a.clip) ? 2.f * e2 : 0.f; }
      }
      if (!rok) { l = 0.f; d = 0.f; }
      dv = d * inv_n;
      if (g == 0) loss_sum += l;
      if (g == 0) gb3 += dv;
    }
    // ---- B1: dY2 = dv w3 act'(H2); dw3, db2 partials; [batch][hidden] copies to LDS
    __syncthreads();                       // (A) previous tile's dW1 reads are done
#pragma unroll
    for (int m = 0; m < NB; ++m) {
      const f32x4 w3v = *reinterpret_cast<const f32x4*>(Bs + 2 * HID + 16 * m + 4 * g);
      f32x4 d;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float hv = h2[m][i];
        gw3[m][i] += dv * hv;
        d[i] = dv * w3v[i] * act_d<ACT>(hv);
        gb2[m][i] += d[i];
      }
      *reinterpret_cast<f32x4*>(trow_d + 16 * m) = d;
      *reinterpret_cast<f32x4*>(trow_h + 16 * m) = h1[m];
    }
    __syncthreads();                       // (B)
    // ---- dW2[q_a][q_b] += sum_b dY2[b][q_a] H1[b][q_b]   (k = batch: 16 steps of 4 rows)
    {
      f32x2 av[2];
      f32x4 bv[2][2];
      auto ld = [&](int t, int buf) {
        const int bt = 4 * t + g;
        av[buf] = *reinterpret_cast<const f32x2*>(Tdy + bt * TPT + 32 * wave + 2 * c);
        const float* p = Th1 + bt * TPT + 8 * c;
        bv[buf][0] = *reinterpret_cast<const f32x4*>(p);
        bv[buf][1] = *reinterpret_cast<const f32x4*>(p + 4);
      };
      ld(0, 0);
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        if (t + 1 < 16) ld(t + 1, (t + 1) & 1);
        fence_sched();
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          const float b = bv[t & 1][n >> 2][n & 3];
          gW2[0][n] = mfma(av[t & 1].x, b, gW2[0][n]);
          gW2[1][n] = mfma(av[t & 1].y, b, gW2[1][n]);
        }
      }
    }
    // ---- B2: dH1^T = W2^T dY2^T.  Step (m, j): k = h2 unit 16 m + 4 g + j, the
    // 8 output blocks kb at once (A = W2[k][u1(kb, c)]: two float4).
    f32x4 d1[NB];
    {
      f32x4 dy2[NB];
#pragma unroll
      for (int m = 0; m < NB; ++m) dy2[m] = *reinterpret_cast<const f32x4*>(trow_d + 16 * m);
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) d1[kb] = (f32x4){0, 0, 0, 0};
      f32x4 A[2][2];
      const float* wb = W2s + (4 * g) * W2P + 16 * (c & 3) + 4 * (c >> 2);
      auto ld = [&](int st, f32x4* d) {
        const float* p = wb + (16 * (st >> 2) + (st & 3)) * W2P;
        d[0] = *reinterpret_cast<const f32x4*>(p);
        d[1] = *reinterpret_cast<const f32x4*>(p + 64);
      };
      ld(0, A[0]);
#pragma unroll
      for (int st = 0; st < 32; ++st) {
        if (st + 1 < 32) ld(st + 1, A[(st + 1) & 1]);
        fence_sched();
        const float b = dy2[st >> 2][st & 3];
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) d1[kb] = mfma(A[st & 1][kb >> 2][kb & 3], b, d1[kb]);
      }
    }
    // dY1 = dH1 act'(H1)  (own H1 values back from the transpose)
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      const f32x4 hv = *reinterpret_cast<const f32x4*>(trow_h + 16 * kb);
#pragma unroll
      for (int i = 0; i < 4; ++i) d1[kb][i] *= act_d<ACT>(hv[i]);
    }
    __syncthreads();                       // (C) all waves finished reading Tdy / Th1
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) *reinterpret_cast<f32x4*>(trow_d + 16 * kb) = d1[kb];
    // X tile [64 rows][features] into the (now free) Th1 region: feature f at
    // 4 (f & 15) + (f >> 4); feature D_in = 1 (the column that yields db1)
    {
      float* xs = Th1 + (wave * 16 + c) * XP;
#pragma unroll
      for (int s = 0; s < KPGE; ++s) {
        const int f = KPGE * g + s;
        if (f < din) xs[4 * (f & 15) + (f >> 4)] = xb[s];
      }
      if (g == 0) xs[4 * (din & 15) + (din >> 4)] = rok ? 1.f : 0.f;
    }
    __syncthreads();                       // (D)
    // ---- dW1[q_a][f] += sum_b dY1[b][q_a] X[b][f]   (reads run 2 steps ahead)
    {
      f32x2 av[3];
      f32x4 bv[3];
      auto ld = [&](int t, int buf) {
        const int bt = 4 * t + g;
        av[buf] = *reinterpret_cast<const f32x2*>(Tdy + bt * TPT + 32 * wave + 2 * c);
        bv[buf] = *reinterpret_cast<const f32x4*>(Th1 + bt * XP + 4 * c);
      };
      ld(0, 0);
      ld(1, 1);
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        if (t + 2 < 16) ld(t + 2, (t + 2) % 3);
        fence_sched();
#pragma unroll
        for (int n = 0; n < NCB; ++n) {
          gW1[0][n] = mfma(av[t % 3].x, bv[t % 3][n], gW1[0][n]);
          gW1[1][n] = mfma(av[t % 3].y, bv[t % 3][n], gW1[1][n]);
        }
      }
    }
  }
  if (!BWD || a.partials == nullptr) return;

  // ---- write this workgroup's partial slab: [W1 | b1 | W2 | b2 | w3 | b3 | loss | pad]
  const int P = mlp_num_params(din);
  float* out = a.partials + (int64_t)blockIdx.x * (P + 2);
  float* oW1 = out;
  float* ob1 = oW1 + HID * din;
  float* oW2 = ob1 + HID;
  float* ob2 = oW2 + HID * HID;
  float* ow3 = ob2 + HID;
  float* ob3 = ow3 + HID;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int qa = 32 * wave + 2 * (4 * g + i) + rb;
#pragma unroll
      for (int n = 0; n < NB; ++n) oW2[qa * HID + u1_of(8 * c + n)] = gW2[rb][n][i];
      const int h = u1_of(qa);
#pragma unroll
      for (int n = 0; n < NCB; ++n) {
        const int f = 16 * n + c;
        if (f < din) oW1[h * din + f] = gW1[rb][n][i];
        else if (f == din) ob1[h] = gW1[rb][n][i];
      }
    }
  }
  // bias-like vectors: reduce over the 16 batch lanes (c) in registers, then
  // over the 4 waves through LDS
  __syncthreads();
  float* red = Th1;                        // [2][HID][4 waves]
#pragma unroll
  for (int m = 0; m < NB; ++m)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v2 = gb2[m][i], v3 = gw3[m][i];
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) {
        v2 += __shfl_xor(v2, off, 16);
        v3 += __shfl_xor(v3, off, 16);
      }
      if (c == 0) {
        const int h = 16 * m + 4 * g + i;
        red[(0 * HID + h) * 4 + wave] = v2;
        red[(1 * HID + h) * 4 + wave] = v3;
      }
    }
  __syncthreads();
  for (int e = tid; e < 2 * HID; e += MLP_BT) {
    const float s = red[e * 4] + red[e * 4 + 1] + red[e * 4 + 2] + red[e * 4 + 3];
    const int which = e / HID, h = e - which * HID;
    (which == 0 ? ob2 : ow3)[h] = s;
  }
  // scalars: gb3 and loss live in the g == 0 lanes of every wave
  float s3 = (g == 0) ? gb3 : 0.f, sl = (g == 0) ? loss_sum : 0.f;
  s3 = wave_sum(s3);
  sl = wave_sum(sl);
  __shared__ float sc[8];
  if (lane == 0) { sc[wave] = s3; sc[4 + wave] = sl; }
  __syncthreads();
  if (tid == 0) {
    ob3[0] = sc[0] + sc[1] + sc[2] + sc[3];
    ob3[1] = sc[4] + sc[5] + sc[6] + sc[7];   // sum of squared errors of this WG
    ob3[2] = 0.f;
  }
}

// Optional Adam step fused into the slab reduction (no clipping: the clip factor
// needs the global norm first; then the caller runs tce_adam_flat instead).
struct AdamArgs {
  float *param, *m, *v, *state;          // param == nullptr: gradient only
  float lr, b1, b2, eps, wd, step;       // step = count INCLUDING this update
};

// grad[p] = sum over the workgroup slabs (fixed order: 4 interleaved groups of
// slabs, then the groups); stats[0] = mean loss, stats[1] += |grad|^2 (the caller
// zeroes stats); with ad.param the Adam update of torch.optim.Adam (L2 weight
// decay in the gradient, mprl/rl/agent/abstract_agent.py:62-82) is applied in
// the same pass.
__global__ __launch_bounds__(256) void mlp_finish_kernel(const float* __restrict__ partials,
                                                         int nparts, int P, int64_t R,
                                                         float* __restrict__ grad,
                                                         float* __restrict__ stats, AdamArgs ad) {
  __shared__ float part[4][64];
  __shared__ float red[4];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int p = blockIdx.x * 64 + col;
  float s = 0.f;
  if (p < P + 1) {
    const float* src = partials + p;
#pragma unroll 8
    for (int i = grp; i < nparts; i += 4) s += src[(int64_t)i * (P + 2)];
  }
  part[grp][col] = s;
  __syncthreads();
  float sq = 0.f;
  if (grp == 0 && p < P + 1) {
    const float g0 = (part[0][col] + part[1][col]) + (part[2][col] + part[3][col]);
    if (p < P) {
      grad[p] = g0;
      sq = g0 * g0;
      if (ad.param) {
        const float w = ad.param[p];
        const float g = ad.wd != 0.f ? g0 + ad.wd * w : g0;
        const float mi = ad.b1 * ad.m[p] + (1.f - ad.b1) * g;
        const float vi = ad.b2 * ad.v[p] + (1.f - ad.b2) * g * g;
        ad.m[p] = mi;
        ad.v[p] = vi;
        const float bc1 = 1.f - powf(ad.b1, ad.step), bc2s = sqrtf(1.f - powf(ad.b2, ad.step));
        ad.param[p] = w - (ad.lr / bc1) * mi / (sqrtf(vi) / bc2s + ad.eps);
      }
    } else {
      stats[0] = g0 / (float)R;                                // mean loss
    }
  }
  const float tot = block_sum(sq, red);
  if (threadIdx.x == 0) {
    atomicAdd(&stats[1], tot);
    if (ad.param && blockIdx.x == 0) ad.state[0] = ad.step;
  }
}

template <int KPGE>
constexpr size_t mlp_lds_bytes() {
  return sizeof(float) * ((size_t)HID * W2P + (size_t)HID * (4 * KPGE + 2) + 3 * HID +
                          2 * (size_t)ROWS_PER_TILE * TPT);
}

template <int ACT, bool BWD, int KPGE>
void mlp_launch(const MlpArgs& a, int grid, hipStream_t st) {
  constexpr size_t lds = mlp_lds_bytes<KPGE>();
  static_assert(lds <= 160 * 1024, "LDS budget");
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_critic_kernel<ACT, BWD, KPGE>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((mlp_critic_kernel<ACT, BWD, KPGE>), dim3(grid), dim3(MLP_BT), lds, st, a);
}

template <int ACT>
void mlp_go(bool bwd, const MlpArgs& a, int grid, hipStream_t st) {
  if (a.din <= 24) {
    if (bwd) mlp_launch<ACT, true, 6>(a, grid, st); else mlp_launch<ACT, false, 6>(a, grid, st);
  } else {
    if (bwd) mlp_launch<ACT, true, 10>(a, grid, st); else mlp_launch<ACT, false, 10>(a, grid, st);
  }
}

}  // namespace

extern "C" {

int tce_mlp_critic_hidden(void) { return HID; }
int tce_mlp_critic_grid(void) { return 256; }
int64_t tce_mlp_critic_num_params(int din) { return mlp_num_params(din); }

// Forward (+ backward when partials != NULL) of the D_in -> 128 -> 128 -> 1
// value network over R rows.  act: 0 tanh, 1 relu, 2 leaky_relu, 3 softplus.
// partials: float [tce_mlp_critic_grid()][num_params + 2]; grad: float
// [num_params] in the order W1, b1, W2, b2, w3, b3 (torch Linear layouts);
// stats: float[2] = {mean loss, |grad|^2}, ZEROED BY THE CALLER ([1] accumulates).
// adam_param != NULL: Adam step on (adam_param, adam_m, adam_v) [num_params]
// fused into the gradient reduction (no clipping), adam_step = step count
// including this update, written to adam_state[0].
int tce_mlp_critic_f32(const float* x, int64_t env_stride, int64_t row_stride, int T,
                       int64_t R, int din, const float* w1, const float* b1,
                       const float* w2, const float* b2, const float* w3, const float* b3,
                       int act, const float* returns, const float* old_values, float clip,
                       float* values, float* partials, float* grad, float* stats,
                       int max_workgroups, float* adam_param, float* adam_m, float* adam_v,
                       float* adam_state, float lr, float beta1, float beta2, float eps,
                       float weight_decay, float adam_step, void* stream) {
  TCE_CHECK_ARG(x && w1 && b1 && w2 && b2 && w3 && b3 && R > 0 && T > 0,
                "mlp_critic: null buffer / bad sizes");
  TCE_CHECK_ARG(din >= 1 && din <= MAX_DIN, "mlp_critic: 1 <= D_in <= 40");
  TCE_CHECK_ARG(act >= 0 && act <= 3, "mlp_critic: unknown activation");
  const bool bwd = partials != nullptr;
  TCE_CHECK_ARG(!bwd || (returns && grad && stats), "mlp_critic: backward buffers missing");
  TCE_CHECK_ARG(bwd || values, "mlp_critic: nothing to compute");
  TCE_CHECK_ARG(!(bwd && clip > 0.f && !old_values), "mlp_critic: old values missing");
  TCE_CHECK_ARG(!adam_param || (bwd && adam_m && adam_v && adam_state && adam_step >= 1.f),
                "mlp_critic: fused Adam needs the backward pass and its state buffers");
  MlpArgs a{x, env_stride, row_stride, T, R, din, w1, b1, w2, b2, w3, b3,
            returns, old_values, clip, values, partials};
  hipStream_t st = (hipStream_t)stream;
  const int64_t ntiles = ceil_div(R, ROWS_PER_TILE);
  int cap = tce_mlp_critic_grid();
  if (max_workgroups > 0 && max_workgroups < cap) cap = max_workgroups;
  const int grid = (int)tmin<int64_t>(cap, ntiles);
  switch (act) {
    case 0: mlp_go<ACT_TANH>(bwd, a, grid, st); break;
    case 1: mlp_go<ACT_RELU>(bwd, a, grid, st); break;
    case 2: mlp_go<ACT_LEAKY>(bwd, a, grid, st); break;
    default: mlp_go<ACT_SOFTPLUS>(bwd, a, grid, st); break;
  }
  TCE_LAUNCH_CHECK();
  if (bwd) {
    const int P = mlp_num_params(din);
    AdamArgs ad{adam_param, adam_m, adam_v, adam_state, lr, beta1, beta2, eps, weight_decay,
                adam_step};
    hipLaunchKernelGGL(mlp_finish_kernel, dim3((unsigned)ceil_div(P + 1, 64)), dim3(256), 0,
                       st, partials, grid, P, R, grad, stats, ad);
    TCE_LAUNCH_CHECK();
  }
  return 0;
}

}  // extern "C"
