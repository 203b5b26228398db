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
// Work decomposition: 64 batch rows per tile (16 per chain wave), workgroups
// persistent over their share of the tiles.  Forward only: 4 waves.  Forward +
// backward: 8 waves in two roles (see mlp_critic_bwd_kernel); the weight-
// gradient accumulators stay in registers across tiles, one partial slab per
// workgroup at the end, reduced (and fed to Adam) by mlp_finish_kernel.
// MFMA-bound: 944 MFMAs per SIMD and tile.
#include "mlp_shared.h"

extern "C" int tce_xchg_adam_f32(void* xchg, float* param, float* grad, float* m, float* v,
                                 int64_t n, float* state, float* norms_out, float step, float lr,
                                 float beta1, float beta2, float eps, float weight_decay,
                                 float clip, float grad_scale, void* stream);
extern "C" int tce_adam_once_f32(float* param, const float* grad, float* m, float* v, int64_t n,
                                 float* state, float* norms_out, float step, float lr,
                                 float beta1, float beta2, float eps, float weight_decay,
                                 float clip, float grad_scale, void* stream);

namespace {

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

// LDS carve shared by both kernels.  KPGE: input features per lane group
// (even); D_in <= 4 KPGE.
template <int KPGE>
struct Lds {
  static constexpr int W1P = 4 * KPGE + 2;                    // even pitch: float2 reads
  float *W2s, *W1s, *Bs, *Th1, *Tdy;
  __device__ explicit Lds(char* raw) {
    W2s = reinterpret_cast<float*>(raw);                      // [HID][W2P]   rows h2, columns h1 unit
    W1s = W2s + HID * W2P;                                    // [HID pos][W1P]
    Bs = W1s + HID * W1P;                                     // b1 (by position) | b2 | w3
    Th1 = Bs + 3 * HID;                                       // [64][TPT] H1 by position; later X stash [64][XP]
    Tdy = Th1 + ROWS_PER_TILE * TPT;                          // [64][TPT] dY2 (by h2), later dY1 (by position)
  }
};

template <int KPGE>
__device__ inline void stage_weights(const MlpArgs& a, const Lds<KPGE>& L, int tid, int nthreads) {
  constexpr int W1P = Lds<KPGE>::W1P;
  const int din = a.din;
  // W2 in 16-byte pieces, a batch of loads in flight per thread: staged element by
  // element (64 dependent-latency round trips per thread in the 256-thread kernel)
  // this loop WAS the policy net's forward on 4096 rows -- 146 us beside the critic
  if ((reinterpret_cast<uintptr_t>(a.w2) & 15) == 0) {
    const f32x4* w2v = reinterpret_cast<const f32x4*>(a.w2);
    constexpr int NV = HID * HID / 4, BATCH = 8;
    for (int e0 = tid; e0 < NV; e0 += BATCH * nthreads) {
      f32x4 v[BATCH];
#pragma unroll
      for (int b = 0; b < BATCH; ++b) {
        const int e = e0 + b * nthreads;
        v[b] = w2v[e < NV ? e : NV - 1];
      }
#pragma unroll
      for (int b = 0; b < BATCH; ++b) {
        const int e = e0 + b * nthreads;
        if (e < NV) *reinterpret_cast<f32x4*>(L.W2s + (e >> 5) * W2P + 4 * (e & 31)) = v[b];
      }
    }
  } else {
    for (int e = tid; e < HID * HID; e += nthreads) L.W2s[(e >> 7) * W2P + (e & 127)] = a.w2[e];
  }
  {
    constexpr int NW1 = HID * 4 * KPGE, BATCH = 8;       // (gathered: position order)
    for (int e0 = tid; e0 < NW1; e0 += BATCH * nthreads) {
      float v[BATCH];
#pragma unroll
      for (int b = 0; b < BATCH; ++b) {
        const int e = e0 + b * nthreads < NW1 ? e0 + b * nthreads : NW1 - 1;
        const int q = e / (4 * KPGE), f = e - q * 4 * KPGE;
        v[b] = a.w1[u1_of(q) * din + (f < din ? f : din - 1)];
      }
#pragma unroll
      for (int b = 0; b < BATCH; ++b) {
        const int e = e0 + b * nthreads;
        if (e < NW1) {
          const int q = e / (4 * KPGE), f = e - q * 4 * KPGE;
          L.W1s[q * W1P + f] = f < din ? v[b] : 0.f;
        }
      }
    }
  }
  for (int e = tid; e < HID; e += nthreads) {
    L.Bs[e] = a.b1[u1_of(e)];
    L.Bs[HID + e] = a.b2[e];
    L.Bs[2 * HID + e] = a.w3 ? a.w3[e] : 0.f;
  }
}

// X fragment of the cursor's row: lane group g holds X[r][KPGE g + s], s < KPGE
// (clamped addresses, no branches around the loads); returns the clamped row.
template <int KPGE>
__device__ inline int64_t load_x(const MlpArgs& a, const RowCursor& cur, int g, float* dst) {
  const bool in = cur.r < a.R;
  int64_t ne = in ? cur.ne : cur.last_ne;
  int t = in ? cur.t : cur.last_t;
  int64_t phys = in ? cur.r : a.R - 1;
  if (a.row_index) {                       // (uniform) gathered rows of a minibatch
    phys = a.row_index[phys];
    ne = phys / a.T;
    t = (int)(phys - ne * a.T);
  }
  const float* xr = a.x + ne * a.env_stride + t * a.row_stride;
#pragma unroll
  for (int s = 0; s < KPGE; ++s) {
    const int k = KPGE * g + s;
    dst[s] = xr[k < a.din ? k : a.din - 1];
  }
  return phys;
}

// Forward chain of one 16-row slice: h1 = act(W1 x + b1) (by position), h2 =
// act(W2 h1 + b2).
// BIAS_LAST: the accumulators start at zero and the biases are added behind the
// MFMAs, so the first MFMA of a step does not wait for an LDS read.  Measured:
// the forward-only kernel gains 2 % (0.733 -> 0.717 ms at the C2 shape); the
// two-role kernel loses 1.5 % (1.998 -> 2.028 ms: while the chain waits for that
// read its SIMD's gradient wave issues MFMAs it must issue anyway) -- so only
// the forward kernel uses it.
// (Tried: the next tile's row loads between the two layers instead of at the top
// of the tile -- no difference, 2.035 ms either way.)
template <int ACT, int KPGE, bool BIAS_LAST = false>
__device__ inline void forward_chain(const Lds<KPGE>& L, const float* xb, int c, int g,
                                     f32x4* h1, f32x4* h2) {
  constexpr int W1P = Lds<KPGE>::W1P;
  // ---- F2: Y1^T = W1 X^T + b1  (A = W1 rows by position, B = X fragment).
  // Two row blocks at a time: a dependent accumulate needs 40 cycles but the
  // MFMA issues every 32, so every chain is paired with a second one.
  {
    float A[2][2 * KPGE];
    auto ld = [&](int mp, float* d) {
      const float* p = L.W1s + (32 * mp + c) * W1P + KPGE * g;
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
      const f32x4 bl0 = *reinterpret_cast<const f32x4*>(L.Bs + 32 * mp + 4 * g);
      const f32x4 bl1 = *reinterpret_cast<const f32x4*>(L.Bs + 32 * mp + 16 + 4 * g);
      const f32x4 zero4 = {0, 0, 0, 0};
      f32x4 acc0 = BIAS_LAST ? zero4 : bl0, acc1 = BIAS_LAST ? zero4 : bl1;
      const f32x4 bias0 = BIAS_LAST ? bl0 : zero4, bias1 = BIAS_LAST ? bl1 : zero4;
      fence_sched();
#pragma unroll
      for (int s = 0; s < KPGE; ++s) {
        acc0 = mfma(A[mp & 1][s], xb[s], acc0);
        acc1 = mfma(A[mp & 1][KPGE + s], xb[s], acc1);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc0[i] = act_f<ACT>(BIAS_LAST ? acc0[i] + bias0[i] : acc0[i]);
        acc1[i] = act_f<ACT>(BIAS_LAST ? acc1[i] + bias1[i] : acc1[i]);
      }
      h1[2 * mp] = acc0;
      h1[2 * mp + 1] = acc1;
    }
  }
  // ---- F4: Y2^T = W2 H1^T + b2.  Step (mp, j): rows 32 mp + c and + 16, the
  // 8 k-steps kb with B = h1[kb][j] (unit 64 (kb>>2) + 16 j + 4 g + (kb&3)).
  {
    f32x4 A[2][4];
    const float* wb = L.W2s + c * W2P + 4 * g;
    auto ld = [&](int st, f32x4* d) {
      const float* p = wb + (32 * (st >> 2)) * W2P + 16 * (st & 3);
      d[0] = *reinterpret_cast<const f32x4*>(p);
      d[1] = *reinterpret_cast<const f32x4*>(p + 64);
      d[2] = *reinterpret_cast<const f32x4*>(p + 16 * W2P);
      d[3] = *reinterpret_cast<const f32x4*>(p + 16 * W2P + 64);
    };
    ld(0, A[0]);
    f32x4 acc0, acc1, bias0 = {0, 0, 0, 0}, bias1 = {0, 0, 0, 0};
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      const int mp = st >> 2, j = st & 3;
      if (st + 1 < 16) ld(st + 1, A[(st + 1) & 1]);
      if (j == 0) {
        const f32x4 bl0 = *reinterpret_cast<const f32x4*>(L.Bs + HID + 32 * mp + 4 * g);
        const f32x4 bl1 = *reinterpret_cast<const f32x4*>(L.Bs + HID + 32 * mp + 16 + 4 * g);
        if (BIAS_LAST) {
          bias0 = bl0;
          bias1 = bl1;
          acc0 = (f32x4){0, 0, 0, 0};
          acc1 = (f32x4){0, 0, 0, 0};
        } else {
          acc0 = bl0;
          acc1 = bl1;
        }
      }
      fence_sched();
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        acc0 = mfma(A[st & 1][kb >> 2][kb & 3], h1[kb][j], acc0);
        acc1 = mfma(A[st & 1][2 + (kb >> 2)][kb & 3], h1[kb][j], acc1);
      }
      if (j == 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc0[i] = act_f<ACT>(BIAS_LAST ? acc0[i] + bias0[i] : acc0[i]);
          acc1[i] = act_f<ACT>(BIAS_LAST ? acc1[i] + bias1[i] : acc1[i]);
        }
        h2[2 * mp] = acc0;
        h2[2 * mp + 1] = acc1;
      }
    }
  }
}

// v = w3 . H2 (without b3; summed over the 4 lane groups of a column)
template <int KPGE>
__device__ inline float value_head(const Lds<KPGE>& L, int g, const f32x4* h2) {
  float v = 0.f;
#pragma unroll
  for (int m = 0; m < NB; ++m) {
    const f32x4 w3v = *reinterpret_cast<const f32x4*>(L.Bs + 2 * HID + 16 * m + 4 * g);
#pragma unroll
    for (int i = 0; i < 4; ++i) v += w3v[i] * h2[m][i];
  }
  return sum_lane_groups(v);
}

// ---- forward only (rollout values): 4 waves, 64 rows per tile -------------
template <int ACT, int KPGE>
__global__ __launch_bounds__(MLP_BT, 1) void mlp_critic_fwd_kernel(MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const Lds<KPGE> L(smem_raw);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  stage_weights<KPGE>(a, L, tid, MLP_BT);
  __syncthreads();
  const float b3 = a.hout ? 0.f : a.b3[0];
  const int64_t ntiles = (a.R + ROWS_PER_TILE - 1) / ROWS_PER_TILE;
  RowCursor cur(a, blockIdx.x, wave, c);
  float xn[KPGE];
  load_x<KPGE>(a, cur, g, xn);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t r = cur.r;
    const bool rok = r < a.R;
    float xb[KPGE];
#pragma unroll
    for (int s = 0; s < KPGE; ++s) xb[s] = (KPGE * g + s < a.din && rok) ? xn[s] : 0.f;
    cur.advance(a.T);
    load_x<KPGE>(a, cur, g, xn);
    f32x4 h1[NB], h2[NB];
    forward_chain<ACT, KPGE, true>(L, xb, c, g, h1, h2);
    if (a.hout) {                          // hidden mode: H2 row, natural unit order
      if (rok) {
        float* dst = a.hout + r * HID + 4 * g;
#pragma unroll
        for (int m = 0; m < NB; ++m) *reinterpret_cast<f32x4*>(dst + 16 * m) = h2[m];
      }
    } else {
      const float v = value_head<KPGE>(L, g, h2) + b3;
      if (rok && g == 0) a.values[r] = v;
    }
  }
}

// ---- forward + loss + backward: 8 waves, two roles ------------------------
// Waves 0-3 ("chain" waves, one per SIMD) run the register-resident chains of
// their 16 rows: forward, loss, dY2, then dH1 = W2^T dY2 and dY1.  Waves 4-7
// ("gradient" waves, the second wave of each SIMD) own the weight-gradient
// accumulators and contract the [batch][hidden] copies in LDS: dW2, db2 for the
// current tile while the chain waves do dH1, and dW1 (+ db1) of the PREVIOUS
// tile while the chain waves run the next forward.  Two waves per SIMD with
// independent MFMA streams keep the matrix pipe busy through each other's
// stalls (LDS latency, activation / loss VALU work, barrier skew); measured: P3
// runs at 97 % of the MFMA rate, in P1 the gradient waves' burst of 96 MFMAs
// delays the chain by about its own length (the arbiter does not interleave it
// into the chain's bubbles; s_setprio does not change that).  Per tile:
//   P1  chain: forward(i), value, loss, dY2(i) | gradient: dW1(i-1)
//   P2  chain: store H1(i), dY2(i) to LDS      |
//   P3  chain: dH1(i), dY1(i)                  | gradient: dW2(i), db2
//   P4  chain: store dY1(i), X(i) to LDS       |
template <int ACT, int KPGE, bool HIDDEN>
__global__ __launch_bounds__(2 * MLP_BT, 1) void mlp_critic_bwd_kernel(MlpArgs a) {
  constexpr int NCB = (4 * KPGE + 1 + 15) / 16;               // dW1 column blocks incl. the ones column
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const Lds<KPGE> L(smem_raw);
  float* Th1 = L.Th1;
  float* Tdy = L.Tdy;
  const int din = a.din;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int role = tid >> 8;                                  // 0 chain, 1 gradient
  const int wave = (tid >> 6) & 3;                            // row slice / output-row block
  const int c = lane & 15, g = lane >> 4;
  stage_weights<KPGE>(a, L, tid, 2 * MLP_BT);
  __syncthreads();
  const int64_t ntiles = (a.R + ROWS_PER_TILE - 1) / ROWS_PER_TILE;
  const int P = mlp_num_params(din);
  float* out = a.partials + (int64_t)blockIdx.x * (P + 2);
  float* oW1 = out;
  float* ob1 = oW1 + HID * din;
  float* oW2 = ob1 + HID;
  float* ob2 = oW2 + HID * HID;
  float* ow3 = ob2 + HID;
  float* ob3 = ow3 + HID;
  __shared__ float sc[8];
  float* red = Th1;                                           // [HID][4 waves] (after the tile loop)

  // The chain waves carry the critical path: they win the issue arbitration of
  // their SIMD, the gradient waves' MFMAs fill the slots they leave free.
  if (role == 0) {
    // ======================= chain waves =======================
    const float b3 = HIDDEN ? 0.f : a.b3[0];
    const float inv_n = 1.f / (float)a.R;
    float gw3[NB][4];
#pragma unroll
    for (int m = 0; m < NB; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) gw3[m][i] = 0.f;
    float gb3 = 0.f, loss_sum = 0.f;
    float* trow_h = Th1 + (wave * 16 + c) * TPT + 4 * g;     // this lane's C/D rows in the transposes
    float* trow_d = Tdy + (wave * 16 + c) * TPT + 4 * g;
    RowCursor cur(a, blockIdx.x, wave, c);
    float xn[KPGE], retn, oldn = 0.f;
    {
      const int64_t rcn = load_x<KPGE>(a, cur, g, xn);
      retn = HIDDEN ? 0.f : a.ret[rcn];
      if (!HIDDEN && a.clip > 0.f) oldn = a.old_v[rcn];
    }
#ifdef MLPX_STAMP
    long long stt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = __builtin_readcyclecounter();
#define STAMP(k) { const long long tn = __builtin_readcyclecounter(); stt[k] += tn - tprev; tprev = tn; }
#else
#define STAMP(k)
#endif
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      STAMP(9)
      const int64_t r = cur.r;                                // this lane's batch row
      const bool rok = r < a.R;
      // ---- P1: forward chain (the next tile's rows are fetched meanwhile)
      float xb[KPGE];
#pragma unroll
      for (int s = 0; s < KPGE; ++s) xb[s] = (KPGE * g + s < din && rok) ? xn[s] : 0.f;
      const float rt = retn, ov = oldn;
      {
        cur.advance(a.T);
        const int64_t rcn = load_x<KPGE>(a, cur, g, xn);
        if (!HIDDEN) {
          retn = a.ret[rcn];
          if (a.clip > 0.f) oldn = a.old_v[rcn];
        }
      }
      f32x4 h1[NB], h2[NB];
      forward_chain<ACT, KPGE>(L, xb, c, g, h1, h2);
      STAMP(0)
      if (HIDDEN) {
        // hidden mode: dY2 = dL/dH2 * act'(H2), the upstream gradient comes from memory
        const float* src = a.gh + (rok ? r : a.R - 1) * HID + 4 * g;
#pragma unroll
        for (int m = 0; m < NB; ++m) {
          const f32x4 gv = *reinterpret_cast<const f32x4*>(src + 16 * m);
#pragma unroll
          for (int i = 0; i < 4; ++i)
            h2[m][i] = rok ? gv[i] * act_d<ACT>(h2[m][i]) : 0.f;
        }
      } else {
        // value, loss and dL/dv (mean over ALL rows R of the epoch), dY2
        const float v = value_head<KPGE>(L, g, h2) + b3;
        if (a.values && rok && g == 0) a.values[r] = v;
        float dv;
        {
          const float e = v - rt;
          float l = e * e, d = 2.f * e;
          if (a.clip > 0.f) {
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
        // dY2 = dv w3 act'(H2) (in place in h2); dw3 partials
  #pragma unroll
        for (int m = 0; m < NB; ++m) {
          const f32x4 w3v = *reinterpret_cast<const f32x4*>(L.Bs + 2 * HID + 16 * m + 4 * g);
  #pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float hv = h2[m][i];
            gw3[m][i] += dv * hv;
            h2[m][i] = dv * w3v[i] * act_d<ACT>(hv);
          }
        }
      }
      STAMP(1)
      __syncthreads();                     // end P1: gradient waves finished dW1(i-1)
      STAMP(2)
      // ---- P2: [batch][hidden] copies of H1 and dY2
#pragma unroll
      for (int m = 0; m < NB; ++m) {
        *reinterpret_cast<f32x4*>(trow_d + 16 * m) = h2[m];
        *reinterpret_cast<f32x4*>(trow_h + 16 * m) = h1[m];
      }
      __syncthreads();                     // end P2
      STAMP(3)
      // ---- P3: dH1^T = W2^T dY2^T.  Step (m, j): k = h2 unit 16 m + 4 g + j,
      // the 8 output blocks kb at once (A = W2[k][u1(kb, c)]: two float4).
      f32x4 d1[NB];
      {
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) d1[kb] = (f32x4){0, 0, 0, 0};
        f32x4 A[2][2];
        const float* wb = L.W2s + (4 * g) * W2P + 16 * (c & 3) + 4 * (c >> 2);
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
          const float b = h2[st >> 2][st & 3];
#pragma unroll
#ifndef MLPX_NOB2
          for (int kb = 0; kb < NB; ++kb) d1[kb] = mfma(A[st & 1][kb >> 2][kb & 3], b, d1[kb]);
#else
          for (int kb = 0; kb < 1; ++kb) d1[st & 7][0] += A[st & 1][0][0] * b + A[st & 1][1][1];
#endif
        }
      }
      STAMP(4)
      // dY1 = dH1 act'(H1)
#pragma unroll
      for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int i = 0; i < 4; ++i) d1[kb][i] *= act_d<ACT>(h1[kb][i]);
      STAMP(5)
      __syncthreads();                     // end P3: gradient waves finished dW2(i)
      STAMP(6)
      // ---- P4: dY1 over dY2; X tile [64 rows][features] over H1: feature f at
      // 4 (f & 15) + (f >> 4) (zeros past D_in); feature D_in = 1 (the column
      // that yields db1; stored after the owning lane's zero: same wave)
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) *reinterpret_cast<f32x4*>(trow_d + 16 * kb) = d1[kb];
      {
        float* xs = Th1 + (wave * 16 + c) * XP;
#pragma unroll
        for (int s = 0; s < KPGE; ++s) {
          const int f = KPGE * g + s;
          xs[4 * (f & 15) + (f >> 4)] = xb[s];
        }
        if (g == 0) xs[4 * (din & 15) + (din >> 4)] = rok ? 1.f : 0.f;
      }
      __syncthreads();                     // end P4
      STAMP(7)
    }
    __syncthreads();                       // gradient waves: dW1 of the last tile
    // ---- dw3: reduce over the 16 batch lanes, then over the 4 chain waves
#pragma unroll
    for (int m = 0; m < NB; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v3 = gw3[m][i];
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) v3 += __shfl_xor(v3, off, 16);
        if (c == 0) red[(16 * m + 4 * g + i) * 4 + wave] = v3;
      }
    float s3 = (g == 0) ? gb3 : 0.f, sl = (g == 0) ? loss_sum : 0.f;
    s3 = wave_sum(s3);
    sl = wave_sum(sl);
    if (lane == 0) { sc[wave] = s3; sc[4 + wave] = sl; }
    __syncthreads();
    if (tid < HID) ow3[tid] = (red[tid * 4] + red[tid * 4 + 1]) + (red[tid * 4 + 2] + red[tid * 4 + 3]);
    if (tid == 0) {
      ob3[0] = sc[0] + sc[1] + sc[2] + sc[3];
      ob3[1] = sc[4] + sc[5] + sc[6] + sc[7];   // sum of squared errors of this WG
      ob3[2] = 0.f;
    }
#ifdef MLPX_STAMP
    __syncthreads();
    if (tid == 0 && blockIdx.x == 0)
      for (int k = 0; k < 10; ++k) out[k] = (float)stt[k];
#endif
  } else {
    // ======================= gradient waves =======================
    f32x4 gW2[2][NB];        // rows q_a = 32 wave + 2 (4 g + i) + rb, columns q_b = 8 c + n
    f32x4 gW1[2][NCB];       // rows as gW2 (positions of hidden 1), columns feature 16 n + c
    float gb2[2] = {0.f, 0.f};                                // unit 32 wave + 2 c + rb, partial over g
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
      for (int n = 0; n < NB; ++n) gW2[rb][n] = (f32x4){0, 0, 0, 0};
#pragma unroll
      for (int n = 0; n < NCB; ++n) gW1[rb][n] = (f32x4){0, 0, 0, 0};
    }
    // dW1[q_a][f] += sum_b dY1[b][q_a] X[b][f]   (reads run 2 steps ahead)
    auto dw1 = [&]() {
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
#ifndef MLPX_NODW1
        for (int n = 0; n < NCB; ++n) {
          gW1[0][n] = mfma(av[t % 3].x, bv[t % 3][n], gW1[0][n]);
          gW1[1][n] = mfma(av[t % 3].y, bv[t % 3][n], gW1[1][n]);
        }
#else
        for (int n = 0; n < 1; ++n) gW1[0][0][0] += av[t % 3].x * bv[t % 3][0];
#endif
      }
    };
    bool first = true;
#ifdef MLPX_STAMP
    long long gst[4] = {0, 0, 0, 0};
#define GSTAMP(k, t0) gst[k] += __builtin_readcyclecounter() - (t0);
#else
#define GSTAMP(k, t0)
#endif
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#ifdef MLPX_STAMP
      const long long tg0 = __builtin_readcyclecounter();
#endif
      if (!first) dw1();                   // P1: previous tile
      first = false;
      GSTAMP(0, tg0)
      __syncthreads();                     // end P1
      GSTAMP(1, tg0)
      __syncthreads();                     // end P2
#ifdef MLPX_STAMP
      const long long tg1 = __builtin_readcyclecounter();
#endif
      // ---- P3: dW2[q_a][q_b] += sum_b dY2[b][q_a] H1[b][q_b]  (k = batch: 16 steps of 4 rows)
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
          gb2[0] += av[t & 1].x;
          gb2[1] += av[t & 1].y;
#pragma unroll
#ifndef MLPX_NODW2
          for (int n = 0; n < NB; ++n) {
            const float b = bv[t & 1][n >> 2][n & 3];
            gW2[0][n] = mfma(av[t & 1].x, b, gW2[0][n]);
            gW2[1][n] = mfma(av[t & 1].y, b, gW2[1][n]);
          }
#else
          for (int n = 0; n < 1; ++n) gW2[0][t & 7][0] += bv[t & 1][0][0] + bv[t & 1][1][1];
#endif
        }
      }
      GSTAMP(2, tg1)
      __syncthreads();                     // end P3
      GSTAMP(3, tg1)
      __syncthreads();                     // end P4
    }
    if (!first) dw1();                     // last tile
    __syncthreads();
    // ---- this workgroup's partial slab: [W1 | b1 | W2 | b2 | w3 | b3 | loss | pad]
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
      float v2 = gb2[rb];
      v2 += __shfl_xor(v2, 16, 64);
      v2 += __shfl_xor(v2, 32, 64);
      if (g == 0) ob2[32 * wave + 2 * c + rb] = v2;
    }
    __syncthreads();                       // matches the chain waves' final barrier
#ifdef MLPX_STAMP
    __syncthreads();
    if (tid == 256 && blockIdx.x == 0)
      for (int k = 0; k < 4; ++k) out[10 + k] = (float)gst[k];
#endif
  }
}

// (forward only: no [batch][hidden] images -- 93 KB instead of 160, so that a
// forward workgroup can share its CU with another small workgroup of the policy
// stream, e.g. the single-workgroup covariance projection that otherwise holds
// the only free CU of a shader engine beside the critic grid)
template <int KPGE>
constexpr size_t mlp_lds_bytes(bool bwd = true) {
  return sizeof(float) * ((size_t)HID * W2P + (size_t)HID * (4 * KPGE + 2) + 3 * HID +
                          (bwd ? 2 * (size_t)ROWS_PER_TILE * TPT : 0));
}

template <int ACT, bool BWD, int KPGE>
void mlp_launch(const MlpArgs& a, int grid, hipStream_t st) {
  constexpr size_t lds = mlp_lds_bytes<KPGE>();
  static_assert(lds <= 160 * 1024, "LDS budget");
  if (BWD && a.gh != nullptr) {
    tce_lds_limit(reinterpret_cast<const void*>(mlp_critic_bwd_kernel<ACT, KPGE, true>), (size_t)(lds));
    hipLaunchKernelGGL((mlp_critic_bwd_kernel<ACT, KPGE, true>), dim3(grid), dim3(2 * MLP_BT), lds,
                       st, a);
  } else if (BWD) {
    tce_lds_limit(reinterpret_cast<const void*>(mlp_critic_bwd_kernel<ACT, KPGE, false>), (size_t)(lds));
    hipLaunchKernelGGL((mlp_critic_bwd_kernel<ACT, KPGE, false>), dim3(grid), dim3(2 * MLP_BT), lds,
                       st, a);
  } else {
    constexpr size_t ldsf = mlp_lds_bytes<KPGE>(false);
    tce_lds_limit(reinterpret_cast<const void*>(mlp_critic_fwd_kernel<ACT, KPGE>), (size_t)(ldsf));
    hipLaunchKernelGGL((mlp_critic_fwd_kernel<ACT, KPGE>), dim3(grid), dim3(MLP_BT), ldsf, st, a);
  }
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

extern "C" int tce_cu_budget_value(void);       // csrc/pair_logprob.hip

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
static int mlp_critic_impl(const float* x, int64_t env_stride, int64_t row_stride, int T,
                           int64_t R, int din, const float* w1, const float* b1,
                           const float* w2, const float* b2, const float* w3, const float* b3,
                           int act, const float* returns, const float* old_values, float clip,
                           float* values, float* partials, float* grad, float* stats,
                           int max_workgroups, float* adam_param, float* adam_m, float* adam_v,
                           float* adam_state, float lr, float beta1, float beta2, float eps,
                           float weight_decay, float adam_step, float grad_scale, void* xchg,
                           void* stream, const int64_t* row_index) {
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
  TCE_CHECK_ARG(!row_index || (bwd && !values),
                "mlp_critic: a row index goes with the backward pass only (no values output)");
  MlpArgs a{x, env_stride, row_stride, T, R, din, w1, b1, w2, b2, w3, b3,
            returns, old_values, clip, values, partials, nullptr, nullptr, row_index};
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
    TCE_CHECK_ARG(!xchg || adam_param, "mlp_critic: an exchange needs the fused Adam step");
    // env shards: the slab reduction leaves the local gradient, the exchange + Adam
    // follow as ONE small launch (few waiting workgroups; csrc/mlp_shared.h)
    AdamArgs ad{xchg ? nullptr : adam_param, adam_m, adam_v, adam_state, lr, beta1, beta2, eps,
                weight_decay, adam_step};
    hipLaunchKernelGGL(mlp_finish_kernel, dim3((unsigned)ceil_div(P + 1, 64)),
                       dim3(64 * FIN_GROUPS), 0, st, partials, grid, P, R, grad, stats, ad);
    TCE_LAUNCH_CHECK();
    if (xchg)
      return tce_xchg_adam_f32(xchg, adam_param, grad, adam_m, adam_v, P, adam_state, stats + 2,
                               adam_step, lr, beta1, beta2, eps, weight_decay, 0.f, grad_scale,
                               stream);
  }
  return 0;
}

int tce_mlp_critic_f32(const float* x, int64_t env_stride, int64_t row_stride, int T,
                       int64_t R, int din, const float* w1, const float* b1,
                       const float* w2, const float* b2, const float* w3, const float* b3,
                       int act, const float* returns, const float* old_values, float clip,
                       float* values, float* partials, float* grad, float* stats,
                       int max_workgroups, float* adam_param, float* adam_m, float* adam_v,
                       float* adam_state, float lr, float beta1, float beta2, float eps,
                       float weight_decay, float adam_step, float grad_scale, void* xchg,
                       void* stream) {
  return mlp_critic_impl(x, env_stride, row_stride, T, R, din, w1, b1, w2, b2, w3, b3, act,
                         returns, old_values, clip, values, partials, grad, stats,
                         max_workgroups, adam_param, adam_m, adam_v, adam_state, lr, beta1, beta2,
                         eps, weight_decay, adam_step, grad_scale, xchg, stream, nullptr);
}

// One critic EPOCH in minibatches (the reference's class default is
// num_minibatchs = 10: mprl/rl/agent/temporal_correlated_agent.py:25,343-366,
// mprl/util/util_data_structure.py:378-391): row_index [R] = the epoch's
// permutation of the rows (the caller draws it with numpy's global generator,
// as the reference does), split like np.array_split into num_minibatches
// consecutive pieces (the first R % k pieces one row longer); per piece ONE
// optimizer step -- forward + value loss (mean over the piece) + backward over
// the gathered rows, slab reduction, Adam -- in the permutation's order.
// stats: float [num_minibatches][4], ZEROED BY THE CALLER: per piece {mean loss,
// |grad|^2, |grad|, |grad| clipped} ([2], [3] only with grad_clip > 0 or an
// exchange; else the caller takes sqrt of [1]).  adam_step: the step count
// INCLUDING the first piece's update; grad_clip: clip_grad_norm (<= 0: none).
int tce_mlp_critic_minibatch_f32(
    const float* x, int64_t env_stride, int64_t row_stride, int T, int64_t R, int din,
    const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
    const float* b3, int act, const float* returns, const float* old_values, float clip,
    const int64_t* row_index, int num_minibatches, float* partials, float* grad, float* stats,
    int max_workgroups, float* adam_param, float* adam_m, float* adam_v, float* adam_state,
    float lr, float beta1, float beta2, float eps, float weight_decay, float adam_step,
    float grad_clip, float grad_scale, void* xchg, void* stream) {
  TCE_CHECK_ARG(row_index && num_minibatches >= 1 && R >= num_minibatches,
                "mlp_critic_minibatch: row index missing / more minibatches than rows");
  TCE_CHECK_ARG(adam_param && adam_m && adam_v && adam_state && partials && grad && stats,
                "mlp_critic_minibatch: optimizer / gradient buffers missing");
  const int P = mlp_num_params(din);
  TCE_CHECK_ARG(!(grad_clip > 0.f) || P <= (1 << 17),
                "mlp_critic_minibatch: clipping needs num_params <= 2^17");
  const bool fused = !(grad_clip > 0.f);
  const int64_t base = R / num_minibatches, extra = R % num_minibatches;
  int64_t off = 0;
  for (int mb = 0; mb < num_minibatches; ++mb) {
    const int64_t len = base + (mb < extra ? 1 : 0);
    float* st4 = stats + 4 * mb;
    const float step = adam_step + (float)mb;
    int rc = mlp_critic_impl(x, env_stride, row_stride, T, len, din, w1, b1, w2, b2, w3, b3, act,
                             returns, old_values, clip, nullptr, partials, grad, st4,
                             max_workgroups, fused ? adam_param : nullptr, adam_m, adam_v,
                             adam_state, lr, beta1, beta2, eps, weight_decay, step, grad_scale,
                             fused ? xchg : nullptr, stream, row_index + off);
    if (rc) return rc;
    if (!fused) {
      rc = xchg ? tce_xchg_adam_f32(xchg, adam_param, grad, adam_m, adam_v, P, adam_state,
                                    st4 + 2, step, lr, beta1, beta2, eps, weight_decay,
                                    grad_clip, grad_scale, stream)
                : tce_adam_once_f32(adam_param, grad, adam_m, adam_v, P, adam_state, st4 + 2,
                                    step, lr, beta1, beta2, eps, weight_decay, grad_clip,
                                    grad_scale, stream);
      if (rc) return rc;
    }
    off += len;
  }
  return 0;
}


// The two hidden layers D_in -> 128 -> 128 of a network with a wider output
// (the policy mean net, mprl/rl/policy/abstract_policy.py:58-99 ->
// mprl/util/util_nn.py:225-246) on the same kernels.  grad_hidden == NULL:
// forward, hidden_out [R][128] = act(W2 act(W1 x + b1) + b2).  Otherwise
// backward of dL/dH2 = grad_hidden [R][128] (the forward is recomputed): grad
// [num_params] receives dW1, db1, dW2, db2 in that order (the trailing w3 / b3
// slots are zero); partials as for tce_mlp_critic_f32; stats: float[2] scratch,
// zeroed by the caller.
int tce_mlp_hidden_f32(const float* x, int64_t env_stride, int64_t row_stride, int T,
                       int64_t R, int din, const float* w1, const float* b1,
                       const float* w2, const float* b2, int act, const float* grad_hidden,
                       float* hidden_out, float* partials, float* grad, float* stats,
                       void* stream) {
  TCE_CHECK_ARG(x && w1 && b1 && w2 && b2 && R > 0 && T > 0, "mlp_hidden: null buffer / bad sizes");
  TCE_CHECK_ARG(din >= 1 && din <= MAX_DIN, "mlp_hidden: 1 <= D_in <= 40");
  TCE_CHECK_ARG(act >= 0 && act <= 3, "mlp_hidden: unknown activation");
  const bool bwd = grad_hidden != nullptr;
  TCE_CHECK_ARG(bwd ? (partials && grad && stats) : (hidden_out != nullptr),
                "mlp_hidden: output buffers missing");
  MlpArgs a{x, env_stride, row_stride, T, R, din, w1, b1, w2, b2, nullptr, nullptr,
            nullptr, nullptr, 0.f, nullptr, bwd ? partials : nullptr,
            bwd ? nullptr : hidden_out, grad_hidden};
  hipStream_t st = (hipStream_t)stream;
  const int64_t ntiles = ceil_div(R, ROWS_PER_TILE);
  int grid = (int)tmin<int64_t>(tce_mlp_critic_grid(), ntiles);
  // beside the critic's persistent grid (tce_set_cu_budget) only that many CUs
  // are free and a workgroup needs a whole one: as many workgroups as CUs, each
  // staging the weights once for several tiles, instead of two or three rounds
  // of one-tile workgroups
#ifndef MLP_HIDDEN_IGNORE_BUDGET
  const int budget = tce_cu_budget_value();
  if (budget > 0 && grid > budget) grid = budget;
#endif
  switch (act) {
    case 0: mlp_go<ACT_TANH>(bwd, a, grid, st); break;
    case 1: mlp_go<ACT_RELU>(bwd, a, grid, st); break;
    case 2: mlp_go<ACT_LEAKY>(bwd, a, grid, st); break;
    default: mlp_go<ACT_SOFTPLUS>(bwd, a, grid, st); break;
  }
  TCE_LAUNCH_CHECK();
  if (bwd) {
    const int P = mlp_num_params(din);
    AdamArgs ad = adam_args_none();
    hipLaunchKernelGGL(mlp_finish_kernel, dim3((unsigned)ceil_div(P + 1, 64)),
                       dim3(64 * FIN_GROUPS), 0, st, partials, grid, P, R, grad, stats, ad);
    TCE_LAUNCH_CHECK();
  }
  return 0;
}

}  // extern "C"
