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
// batch].  An MFMA result tile (batch column on the lane, hidden rows in the 4
// registers) is then directly the B operand of the next layer's MFMA (which
// contracts over the hidden index), so the forward chain X -> H1 -> H2 -> v and
// the backward chain dY2 -> dH1 never leave the accumulator registers; only
// the weights (A operands) come from LDS.  The weight gradients contract over
// the batch index, which sits on the lanes: for those the tiles are written
// once to LDS ([hidden][batch], pitch 17) and re-read as A/B fragments, and
// the 4 waves of a workgroup split the output blocks of dW2 / dW1.
//
// Work decomposition: a workgroup = 4 waves = 64 batch rows per tile (16 per
// wave), persistent over its share of the tiles; weight-gradient accumulators
// stay in registers across tiles; one partial slab per workgroup at the end,
// reduced by mlp_reduce_kernel.  MFMA-bound: 2*(D_in*H + H*H) fwd + about twice
// that backward per row.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HID = 128;
constexpr int NB = HID / 16;          // 8 row blocks of 16 hidden units
constexpr int W2P = HID + 1;          // LDS pitch of W2 (bank-conflict free rows)
constexpr int TP = 17;                // pitch of the [hidden][16 batch] transposes
constexpr int XSP = 49;               // pitch of the X tile stashed for dW1 (>= 4 * MAXKPG, odd)
constexpr int MAXKPG = 12;            // D_in <= 48 (LDS budget)
constexpr int MLP_BT = 256;
constexpr int ROWS_PER_TILE = 64;

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

template <int ACT, bool BWD>
__global__ __launch_bounds__(MLP_BT, 1) void mlp_critic_kernel(MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* W2s = reinterpret_cast<float*>(smem_raw);            // [HID][W2P]
  float* W1s = W2s + HID * W2P;                               // [HID][w1p]
  const int din = a.din;
  const int kpg = (din + 3) >> 2;                             // k per lane group
  const int w1p = (4 * kpg) | 1;                              // odd pitch
  float* Bs = W1s + HID * w1p;                                // [2][HID] biases b1, b2
  float* Th1 = Bs + 2 * HID;                                  // [4 waves][HID][TP]
  float* Tdy = Th1 + 4 * HID * TP;                            // [4 waves][HID][TP]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;

  // ---- stage the weights once per workgroup
  for (int e = tid; e < HID * HID; e += MLP_BT) W2s[(e >> 7) * W2P + (e & 127)] = a.w2[e];
  for (int e = tid; e < HID * 4 * kpg; e += MLP_BT) {
    const int h = e / (4 * kpg), k = e - h * 4 * kpg;
    W1s[h * w1p + k] = k < din ? a.w1[h * din + k] : 0.f;
  }
  for (int e = tid; e < HID; e += MLP_BT) { Bs[e] = a.b1[e]; Bs[HID + e] = a.b2[e]; }
  __syncthreads();

  // per-lane constants: rows owned in a C/D tile of block m: 16 m + 4 g + i
  float w3r[NB][4];
#pragma unroll
  for (int m = 0; m < NB; ++m)
#pragma unroll
    for (int i = 0; i < 4; ++i) w3r[m][i] = a.w3[16 * m + 4 * g + i];
  const float b3 = a.b3[0];

  // persistent gradient accumulators
  f32x4 gW2[2][NB];        // wave owns h2 rows [32 wave, 32 wave + 32) x all h1
  f32x4 gW1[2][3];         // wave owns h1 rows [32 wave, +32) x din (<= 48 -> 3 col blocks)
  float gb1[NB][4], gb2[NB][4], gw3[NB][4];
  float gb3 = 0.f, loss_sum = 0.f;
  if (BWD) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
      for (int n = 0; n < NB; ++n) gW2[r][n] = (f32x4){0, 0, 0, 0};
#pragma unroll
      for (int n = 0; n < 3; ++n) gW1[r][n] = (f32x4){0, 0, 0, 0};
    }
#pragma unroll
    for (int m = 0; m < NB; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) { gb1[m][i] = 0.f; gb2[m][i] = 0.f; gw3[m][i] = 0.f; }
  }

  const int64_t ntiles = (a.R + ROWS_PER_TILE - 1) / ROWS_PER_TILE;
  const float inv_n = 1.f / (float)a.R;
  float* th1 = Th1 + wave * HID * TP;
  float* tdy = Tdy + wave * HID * TP;

  // X fragment of a tile: lane (c, g) holds X[r][kpg*g + s], s < kpg.  Loaded
  // one tile ahead (clamped addresses, no branches around the loads) so that
  // the HBM latency hides behind the previous tile's MFMAs.
  auto load_x = [&](int64_t tile, float* dst) {
    const int64_t rr = tile * ROWS_PER_TILE + wave * 16 + c;
    const int64_t rcl = rr < a.R ? rr : a.R - 1;
    const int64_t ne = rcl / a.T;
    const float* xr = a.x + ne * a.env_stride + (rcl - ne * a.T) * a.row_stride;
#pragma unroll
    for (int s = 0; s < MAXKPG; ++s) {
      const int k = kpg * g + s;
      dst[s] = xr[k < din ? k : din - 1];
    }
  };
  float xn[MAXKPG];
  if ((int64_t)blockIdx.x < ntiles) load_x(blockIdx.x, xn);

  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * ROWS_PER_TILE;
    const int64_t r = row0 + wave * 16 + c;                   // this lane's batch row
    const bool rok = r < a.R;
    const int64_t rc = rok ? r : a.R - 1;

    // ---- F1: take the prefetched fragment, start the next tile's loads
    float xb[MAXKPG];
#pragma unroll
    for (int s = 0; s < MAXKPG; ++s) {
      const int k = kpg * g + s;
      xb[s] = (s < kpg && k < din && rok) ? xn[s] : 0.f;
    }
    {
      const int64_t nt = tile + gridDim.x;
      load_x(nt < ntiles ? nt : tile, xn);
    }
    // ---- F2: Y1^T = W1 X^T + b1  (A = W1 from LDS, B = X fragment).  Two row
    // blocks at a time: v_mfma_f32_16x16x4_f32 issues every 32 cycles but a
    // dependent accumulate needs 40, so every chain is paired with a second one.
    f32x4 h1[NB];
#pragma unroll
    for (int m = 0; m < NB; m += 2) {
      const float* bb1 = Bs + 16 * m + 4 * g;
      f32x4 acc0 = {bb1[0], bb1[1], bb1[2], bb1[3]};
      f32x4 acc1 = {bb1[16], bb1[17], bb1[18], bb1[19]};
      const float* wr0 = W1s + (16 * m + c) * w1p + kpg * g;
      const float* wr1 = wr0 + 16 * w1p;
#pragma unroll
      for (int s = 0; s < MAXKPG; ++s)
        if (s < kpg) {
          acc0 = mfma(wr0[s], xb[s], acc0);
          acc1 = mfma(wr1[s], xb[s], acc1);
        }
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc0[i] = act_f<ACT>(acc0[i]); acc1[i] = act_f<ACT>(acc1[i]); }
      h1[m] = acc0;
      h1[m + 1] = acc1;
    }
    // ---- F4: Y2^T = W2 H1^T + b2  (B = H1 accumulators, k = 16 kb + 4 g + j)
    f32x4 h2[NB];
#pragma unroll
    for (int m = 0; m < NB; m += 2) {
      const float* bb2 = Bs + HID + 16 * m + 4 * g;
      f32x4 acc0 = {bb2[0], bb2[1], bb2[2], bb2[3]};
      f32x4 acc1 = {bb2[16], bb2[17], bb2[18], bb2[19]};
      const float* wr0 = W2s + (16 * m + c) * W2P + 4 * g;
      const float* wr1 = wr0 + 16 * W2P;
#pragma unroll
      for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc0 = mfma(wr0[16 * kb + j], h1[kb][j], acc0);
          acc1 = mfma(wr1[16 * kb + j], h1[kb][j], acc1);
        }
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc0[i] = act_f<ACT>(acc0[i]); acc1[i] = act_f<ACT>(acc1[i]); }
      h2[m] = acc0;
      h2[m + 1] = acc1;
    }
    // ---- F5: v = w3 . H2 + b3 (sum over the 4 lane groups of a column)
    float v = 0.f;
#pragma unroll
    for (int m = 0; m < NB; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) v += w3r[m][i] * h2[m][i];
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
    // ---- B1: dY2 = dv w3 act'(H2); dw3, db2 partials; transposes to LDS
    __syncthreads();                       // previous tile's dW1 reads are done
#pragma unroll
    for (int m = 0; m < NB; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float hv = h2[m][i];
        gw3[m][i] += dv * hv;
        const float d = dv * w3r[m][i] * act_d<ACT>(hv);
        gb2[m][i] += d;
        h2[m][i] = d;                      // h2 now holds dY2^T
        const int h = 16 * m + 4 * g + i;
        tdy[h * TP + c] = d;
        th1[h * TP + c] = h1[m][i];
      }
    __syncthreads();
    // ---- dW2[h2][h1] += sum_b dY2^T[h2][b] H1^T[h1][b]  (k = batch: 16 steps)
#pragma unroll 4
    for (int t = 0; t < 16; ++t) {
      const int ws = t >> 2;               // source wave of batch rows 4t..4t+3
      const int cb = 4 * (t & 3) + g;      // batch column inside that wave
      const float* sd = Tdy + ws * HID * TP + cb;
      const float* sh = Th1 + ws * HID * TP + cb;
      const float a0 = sd[(32 * wave + c) * TP];
      const float a1 = sd[(32 * wave + 16 + c) * TP];
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        const float b = sh[(16 * n + c) * TP];
        gW2[0][n] = mfma(a0, b, gW2[0][n]);
        gW2[1][n] = mfma(a1, b, gW2[1][n]);
      }
    }
    // ---- B2: dH1^T = W2^T dY2^T, dY1 = dH1 act'(H1)   (two chains at a time)
    f32x4 d1[NB];
#pragma unroll
    for (int kb = 0; kb < NB; kb += 2) {
      f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
      for (int m = 0; m < NB; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float* wr = W2s + (16 * m + 4 * g + j) * W2P + 16 * kb + c;
          acc0 = mfma(wr[0], h2[m][j], acc0);
          acc1 = mfma(wr[16], h2[m][j], acc1);
        }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float da = acc0[i] * act_d<ACT>(h1[kb][i]);
        const float db = acc1[i] * act_d<ACT>(h1[kb + 1][i]);
        acc0[i] = da;
        acc1[i] = db;
        gb1[kb][i] += da;
        gb1[kb + 1][i] += db;
      }
      d1[kb] = acc0;
      d1[kb + 1] = acc1;
    }
    __syncthreads();                       // all waves finished reading Tdy / Th1
#pragma unroll
    for (int m = 0; m < NB; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) tdy[(16 * m + 4 * g + i) * TP + c] = d1[m][i];
    // X tile [64 rows][din] into the (now free) Th1 region for dW1's B operand
    {
      float* xs = Th1 + (wave * 16 + c) * XSP + kpg * g;
#pragma unroll
      for (int s = 0; s < MAXKPG; ++s)
        if (s < kpg) xs[s] = xb[s];
    }
    __syncthreads();
    // ---- dW1[h1][in] += sum_b dY1^T[h1][b] X[b][in]   (B straight from global / L1)
#pragma unroll 2
    for (int t = 0; t < 16; ++t) {
      const int ws = t >> 2;
      const int cb = 4 * (t & 3) + g;
      const float* sd = Tdy + ws * HID * TP + cb;
      const float a0 = sd[(32 * wave + c) * TP];
      const float a1 = sd[(32 * wave + 16 + c) * TP];
      const float* xr = Th1 + (4 * t + g) * XSP;              // X row of this k (zeros past R / din)
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        if (16 * n < din) {                                   // uniform
          const int k = 16 * n + c;
          const float b = k < 4 * kpg ? xr[k] : 0.f;
          gW1[0][n] = mfma(a0, b, gW1[0][n]);
          gW1[1][n] = mfma(a1, b, gW1[1][n]);
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
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        oW2[(32 * wave + 16 * rb + 4 * g + i) * HID + 16 * n + c] = gW2[rb][n][i];
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = 16 * n + c;
        if (k < din) oW1[(32 * wave + 16 * rb + 4 * g + i) * din + k] = gW1[rb][n][i];
      }
  }
  // bias-like vectors: reduce over the 16 batch lanes (c) in registers, then
  // over the 4 waves through LDS
  __syncthreads();
  float* red = Th1;                        // [3][HID][4 waves]
#pragma unroll
  for (int m = 0; m < NB; ++m)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v1 = gb1[m][i], v2 = gb2[m][i], v3 = gw3[m][i];
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) {
        v1 += __shfl_xor(v1, off, 16);
        v2 += __shfl_xor(v2, off, 16);
        v3 += __shfl_xor(v3, off, 16);
      }
      if (c == 0) {
        const int h = 16 * m + 4 * g + i;
        red[(0 * HID + h) * 4 + wave] = v1;
        red[(1 * HID + h) * 4 + wave] = v2;
        red[(2 * HID + h) * 4 + wave] = v3;
      }
    }
  __syncthreads();
  for (int e = tid; e < 3 * HID; e += MLP_BT) {
    const float s = red[e * 4] + red[e * 4 + 1] + red[e * 4 + 2] + red[e * 4 + 3];
    const int which = e / HID, h = e - which * HID;
    (which == 0 ? ob1 : which == 1 ? ob2 : ow3)[h] = s;
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

// grad[p] = sum over workgroups; out_stats[0] = mean loss, [1] = |grad|^2
__global__ __launch_bounds__(256) void mlp_reduce_kernel(const float* __restrict__ partials,
                                                         int nparts, int P, int64_t R,
                                                         float* __restrict__ grad,
                                                         float* __restrict__ stats) {
  __shared__ float red[4];
  float sq = 0.f;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < P + 1; p += gridDim.x * 256) {
    float s = 0.f;
    for (int i = 0; i < nparts; ++i) s += partials[(int64_t)i * (P + 2) + p];
    if (p < P) { grad[p] = s; sq += s * s; }
    else stats[0] = s / (float)R;                              // mean loss
  }
  const float tot = block_sum(sq, red);
  if (threadIdx.x == 0) atomicAdd(&stats[1], tot);
}

size_t mlp_lds_bytes(int din) {
  const int kpg = (din + 3) >> 2;
  const int w1p = (4 * kpg) | 1;
  const size_t fl = (size_t)HID * W2P + (size_t)HID * w1p + 2 * HID + 2 * 4 * (size_t)HID * TP;
  return fl * sizeof(float);
}

template <int ACT>
int mlp_go(bool bwd, const MlpArgs& a, int grid, hipStream_t st) {
  const size_t lds = mlp_lds_bytes(a.din);
  if (bwd) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_critic_kernel<ACT, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((mlp_critic_kernel<ACT, true>), dim3(grid), dim3(MLP_BT), lds, st, a);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_critic_kernel<ACT, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((mlp_critic_kernel<ACT, false>), dim3(grid), dim3(MLP_BT), lds, st, a);
  }
  return 0;
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
// stats: float[2] = {mean loss, |grad|^2} (zeroed by the call).
int tce_mlp_critic_f32(const float* x, int64_t env_stride, int64_t row_stride, int T,
                       int64_t R, int din, const float* w1, const float* b1,
                       const float* w2, const float* b2, const float* w3, const float* b3,
                       int act, const float* returns, const float* old_values, float clip,
                       float* values, float* partials, float* grad, float* stats,
                       int max_workgroups, void* stream) {
  TCE_CHECK_ARG(x && w1 && b1 && w2 && b2 && w3 && b3 && R > 0 && T > 0,
                "mlp_critic: null buffer / bad sizes");
  TCE_CHECK_ARG(din >= 1 && din <= 48, "mlp_critic: 1 <= D_in <= 48");
  TCE_CHECK_ARG(act >= 0 && act <= 3, "mlp_critic: unknown activation");
  const bool bwd = partials != nullptr;
  TCE_CHECK_ARG(!bwd || (returns && grad && stats), "mlp_critic: backward buffers missing");
  TCE_CHECK_ARG(bwd || values, "mlp_critic: nothing to compute");
  TCE_CHECK_ARG(!(bwd && clip > 0.f && !old_values), "mlp_critic: old values missing");
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
    (void)hipMemsetAsync(stats, 0, 2 * sizeof(float), st);
    hipLaunchKernelGGL(mlp_reduce_kernel, dim3((unsigned)ceil_div(P + 1, 256)), dim3(256), 0,
                       st, partials, grid, P, R, grad, stats);
    TCE_LAUNCH_CHECK();
  }
  return 0;
}

}  // extern "C"
