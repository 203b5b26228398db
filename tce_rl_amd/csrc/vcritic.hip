// One full-batch epoch of the black-box agent's value function on the row kernels
// of csrc/pmlp.hip, for the critics the 64-wide row kernels of csrc/smlp.hip and
// the 256 x 2 kernels of csrc/mlpw_impl.h do not cover -- table tennis's 256 x 1
// (mprl/config/table_tennis_4d/bbrl/entire/shared.yaml:90-91):
//
//   values_new = critic(states)                        black_box_agent.py:128-131
//   loss = value_loss(values_new, returns, old_vs)     :438-466
//   loss.backward(); grad_norm_clip; Adam step         :135-146
//
// without autograd and without a library GEMM: pmlp forward (K = 1), the value
// loss and its gradient per row (one kernel, the loss summed in a fixed order),
// pmlp backward into the flat gradient, clip + Adam as one launch.
#include "common.h"
#include "../../include/tce_hip.h"

namespace {

constexpr int VL_BT = 256, VL_MAX_BLOCKS = 256;

// g[i] = d loss / d values[i], loss = mean(max((ret - v)^2, (clip(v) - ret)^2));
// part[b] = this block's share of the sum; the last block to finish adds the
// shares in order and writes the mean (ticket re-armed for the next launch).
template <typename real>
__global__ __launch_bounds__(VL_BT) void value_loss_kernel(
    const real* __restrict__ values, const real* __restrict__ returns,
    const real* __restrict__ old_values, int64_t N, real clip, real* __restrict__ g,
    double* __restrict__ part, unsigned* __restrict__ ticket, real* __restrict__ loss_out) {
  __shared__ double red[VL_BT / 64];
  __shared__ int last_s;
  const real inv_n = real(1) / (real)N;
  double acc = 0;
  for (int64_t i = blockIdx.x * (int64_t)VL_BT + threadIdx.x; i < N;
       i += (int64_t)gridDim.x * VL_BT) {
    const real v = values[i], r = returns[i];
    real d = v - r;
    real l = d * d;
    real gi = real(2) * d;
    if (clip > real(0)) {
      const real o = old_values[i];
      const real dv = v - o;
      const real dc = tmin(tmax(dv, -clip), clip);
      const real e = o + dc - r;
      const real lc = e * e;
      // torch.max hands the gradient to the larger branch and splits it evenly
      // on a tie; the clamp passes it inside its range (bounds included)
      const real ge = (dv >= -clip && dv <= clip) ? e : real(0);
      if (lc > l) {
        l = lc;
        gi = real(2) * ge;
      } else if (lc == l) {
        gi = d + ge;
      }
    }
    g[i] = gi * inv_n;
    acc += (double)l;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = acc;
    __threadfence();
    last_s = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last_s || threadIdx.x != 0) return;
  __threadfence();
  double s = 0;
  for (unsigned b = 0; b < gridDim.x; ++b) s += __hip_atomic_load(part + b, __ATOMIC_RELAXED,
                                                                  __HIP_MEMORY_SCOPE_AGENT);
  loss_out[0] = (real)(s / (double)N);
  *ticket = 0u;
}

template <typename real> struct VcApi;
template <> struct VcApi<float> {
  static constexpr auto fwd = tce_pmlp_forward_f32;
  static constexpr auto bwd = tce_pmlp_backward_f32;
  static constexpr auto adam_once = tce_adam_once_f32;
  static constexpr auto xadam = tce_xchg_adam_f32;
};
template <> struct VcApi<double> {
  static constexpr auto fwd = tce_pmlp_forward_f64;
  static constexpr auto bwd = tce_pmlp_backward_f64;
  static constexpr auto adam_once = tce_adam_once_f64;
  static constexpr auto xadam = tce_xchg_adam_f64;
};

inline int64_t vc_up4(int64_t n) { return (n + 3) / 4 * 4; }

template <typename real>
int critic_epoch(const real* x, int64_t x_stride, const real* returns, const real* old_values,
                 int64_t N, int din, int H, int NL, int act, real clip_critic, real* param,
                 real* grad, real* m, real* v, real* opt_state, real lr, real beta1, real beta2,
                 real eps, real weight_decay, real clip_grad, real grad_scale, int do_adam,
                 real step, real* ws, real* partials, real* rec_row3, void* xchg, void* stream) {
  typedef VcApi<real> A;
  TCE_CHECK_ARG(!xchg || do_adam, "pmlp_critic_epoch: an exchange needs the Adam step");
  TCE_CHECK_ARG(x && returns && param && grad && ws && partials && rec_row3 && N > 0,
                "pmlp_critic_epoch: null buffer / no rows");
  TCE_CHECK_ARG(clip_critic <= real(0) || old_values,
                "pmlp_critic_epoch: clip_critic > 0 needs the old values");
  TCE_CHECK_ARG(tce_pmlp_supported(din, H, NL, 1, (int)sizeof(real)),
                "pmlp_critic_epoch: net shape not built (tce_pmlp_supported)");
  TCE_CHECK_ARG(!do_adam || (m && v && opt_state), "pmlp_critic_epoch: optimizer state missing");
  const int64_t P = tce_pmlp_num_params(din, H, NL, 1);
  TCE_CHECK_ARG(!do_adam || P <= (1 << 17), "pmlp_critic_epoch: more than 2^17 parameters");
  // ws: h1 [N,H] | h2 [N,H] | values [N] | g [N] | loss partials (double) | ticket
  real* h1 = ws;
  real* h2 = h1 + vc_up4(N * (int64_t)H);
  real* val = h2 + vc_up4(N * (int64_t)H);
  real* g = val + vc_up4(N);
  double* part = reinterpret_cast<double*>(g + vc_up4(N));
  unsigned* ticket = reinterpret_cast<unsigned*>(part + VL_MAX_BLOCKS);
  hipStream_t st = (hipStream_t)stream;
  int rc = A::fwd(x, x_stride, N, din, H, NL, 1, act, param, h1, NL == 2 ? h2 : nullptr, val,
                  stream);
  if (rc) return rc;
  const unsigned grid = (unsigned)tmin<int64_t>(ceil_div(N, VL_BT), VL_MAX_BLOCKS);
  hipLaunchKernelGGL(value_loss_kernel<real>, dim3(grid), dim3(VL_BT), 0, st, (const real*)val,
                     returns, old_values, N, clip_critic, g, part, ticket, rec_row3);
  TCE_LAUNCH_CHECK();
  rc = A::bwd(x, x_stride, N, din, H, NL, 1, act, param, h1, NL == 2 ? h2 : nullptr, g, partials,
              grad, stream);
  if (rc || !do_adam) return rc;
  if (xchg)       // env shards: the peers' gradients are added inside the Adam launch
    return A::xadam(xchg, param, grad, m, v, P, opt_state, rec_row3 + 1, step, lr, beta1, beta2,
                    eps, weight_decay, clip_grad, grad_scale, stream);
  return A::adam_once(param, grad, m, v, P, opt_state, rec_row3 + 1, step, lr, beta1, beta2, eps,
                      weight_decay, clip_grad, grad_scale, stream);
}

}  // namespace

extern "C" {

int64_t tce_pmlp_critic_ws_len(int64_t N, int hidden) {
  // (elements of the net's type; the tail holds VL_MAX_BLOCKS doubles + a ticket)
  return 2 * vc_up4(N * (int64_t)hidden) + 2 * vc_up4(N) + 2 * VL_MAX_BLOCKS + 4;
}

int tce_pmlp_critic_epoch_f32(const float* x, int64_t x_stride, const float* returns,
                              const float* old_values, int64_t N, int din, int hidden,
                              int num_hidden, int act, float clip_critic, float* param, float* grad,
                              float* m, float* v, float* opt_state, float lr, float beta1,
                              float beta2, float eps, float weight_decay, float clip_grad,
                              float grad_scale, int do_adam, float step, float* ws,
                              float* partials, float* rec_row3, void* xchg, void* stream) {
  return critic_epoch<float>(x, x_stride, returns, old_values, N, din, hidden, num_hidden, act,
                             clip_critic, param, grad, m, v, opt_state, lr, beta1, beta2, eps,
                             weight_decay, clip_grad, grad_scale, do_adam, step, ws, partials,
                             rec_row3, xchg, stream);
}
int tce_pmlp_critic_epoch_f64(const double* x, int64_t x_stride, const double* returns,
                              const double* old_values, int64_t N, int din, int hidden,
                              int num_hidden, int act, double clip_critic, double* param,
                              double* grad, double* m, double* v, double* opt_state, double lr,
                              double beta1, double beta2, double eps, double weight_decay,
                              double clip_grad, double grad_scale, int do_adam, double step,
                              double* ws, double* partials, double* rec_row3, void* xchg,
                              void* stream) {
  return critic_epoch<double>(x, x_stride, returns, old_values, N, din, hidden, num_hidden, act,
                              clip_critic, param, grad, m, v, opt_state, lr, beta1, beta2, eps,
                              weight_decay, clip_grad, grad_scale, do_adam, step, ws, partials,
                              rec_row3, xchg, stream);
}

}  // extern "C"
