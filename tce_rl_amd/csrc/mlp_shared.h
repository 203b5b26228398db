// Pieces shared by the fused critic kernels (mlp.hip: exact-fp32 matrix cores,
// mlp16.hip: split-f16 matrix cores): argument block, row cursor, activations,
// the gradient-slab reduction + Adam kernel.
#pragma once
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

// v summed over the lanes l, l ^ 16, l ^ 32, l ^ 48 (all four get the total):
// two VALU swaps (v_permlane16_swap / v_permlane32_swap, gfx950) instead of two
// LDS-routed shuffles.
__device__ inline float sum_lane_groups(float v) {
  {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  return v;
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
  // "hidden" mode (the two hidden layers of a wider-output net, e.g. the policy
  // mean net): forward stores H2 [R][HID] instead of the value, backward takes
  // dL/dH2 [R][HID] instead of the value loss; w3 / b3 are not used.
  float* hout;
  const float* gh;
  // minibatch: logical row r of this launch is row row_index[r] of x / ret /
  // old_v (generate_minibatches + select_batch,
  // mprl/util/util_data_structure.py:362-391); nullptr: rows in place
  const int64_t* row_index;
};

__host__ __device__ inline int mlp_num_params(int din) {
  return HID * din + HID + HID * HID + HID + HID + 1;
}

// This lane's batch row r = tile * 64 + 16 w + c as (env, step), advanced by
// gridDim.x tiles at a time without a division in the tile loop.
struct RowCursor {
  int64_t r, ne;        // row index (unclamped), env
  int t;                // step inside the env
  int64_t dn;           // per advance: envs
  int dt;               //              steps
  int64_t dr;           //              rows
  int64_t last_ne;      // (R - 1) as (env, step): rows past the end read this one
  int last_t;
  __device__ RowCursor(const MlpArgs& a, int64_t tile, int w, int c) {
    r = tile * ROWS_PER_TILE + w * 16 + c;
    ne = r / a.T;
    t = (int)(r - ne * a.T);
    dr = (int64_t)gridDim.x * ROWS_PER_TILE;
    dn = dr / a.T;
    dt = (int)(dr - dn * a.T);
    last_ne = (a.R - 1) / a.T;
    last_t = (int)((a.R - 1) - last_ne * a.T);
  }
  __device__ void advance(int T) {
    r += dr; ne += dn; t += dt;
    if (t >= T) { t -= T; ++ne; }
  }
};

// Optional Adam step fused into the slab reduction (no clipping: the clip factor
// needs the global norm first; then the caller runs tce_adam_flat instead).
struct AdamArgs {
  float *param, *m, *v, *state;          // param == nullptr: gradient only
  float lr, b1, b2, eps, wd, step;       // step = count INCLUDING this update
};
inline AdamArgs adam_args_none() {
  return AdamArgs{nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
}

// grad[p] = sum over the workgroup slabs (fixed order: FIN_GROUPS interleaved
// groups of slabs, then the groups); stats[0] = mean loss, stats[1] += |grad|^2 (the caller
// zeroes stats); with ad.param the Adam update of torch.optim.Adam (L2 weight
// decay in the gradient, mprl/rl/agent/abstract_agent.py:62-82) is applied in
// the same pass.  (Env shards: gradient only here -- this grid is hundreds of
// workgroups, and workgroups that WAIT for a peer must be few: they hold their
// compute units while they wait -- and the exchange + Adam follow as one small
// launch, tce_xchg_adam_*.)
constexpr int FIN_GROUPS = 16;          // slab groups summed in parallel per column

__global__ __launch_bounds__(64 * FIN_GROUPS) void mlp_finish_kernel(
    const float* __restrict__ partials, int nparts, int P, int64_t R, float* __restrict__ grad,
    float* __restrict__ stats, AdamArgs ad) {
  __shared__ float part[FIN_GROUPS][64];
  __shared__ float red[FIN_GROUPS];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int p = blockIdx.x * 64 + col;
  float s = 0.f;
  if (p < P + 1) {
    const float* src = partials + p;
#pragma unroll 4
    for (int i = grp; i < nparts; i += FIN_GROUPS) s += src[(int64_t)i * (P + 2)];
  }
  part[grp][col] = s;
  __syncthreads();
  float sq = 0.f;
  if (grp == 0 && p < P + 1) {
    float g0 = 0.f;
#pragma unroll
    for (int k = 0; k < FIN_GROUPS; ++k) g0 += part[k][col];   // fixed order
    if (p < P) {
      grad[p] = g0;
      sq = g0 * g0;
      if (ad.param) {
        float w = ad.param[p], mi = ad.m[p], vi = ad.v[p], step_size, bc2s;
        adam_coef(ad.lr, ad.b1, ad.b2, ad.step, step_size, bc2s);
        adam_elem(g0, w, mi, vi, ad.b1, ad.b2, ad.eps, ad.wd, step_size, bc2s);
        ad.m[p] = mi;
        ad.v[p] = vi;
        ad.param[p] = w;
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

}  // namespace
