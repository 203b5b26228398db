// Flat Adam step with global-norm gradient clipping, one parameter buffer per
// network.  Replaces, per optimizer step of the update loops
// (mprl/rl/agent/temporal_correlated_agent.py:361-366,597-612):
//   grad_norm_clip(bound, params)            mprl/util/util_numerical.py:244-275
//   torch.optim.Adam(lr, weight_decay).step  mprl/rl/agent/abstract_agent.py:62-82
// The step count, the two gradient norms and the clip factor live in a small
// device state vector so that nothing here needs the host (the update loops are
// replayed from a HIP graph).  HBM-bound elementwise work: 7 reads/writes of
// the element type per parameter.
#include "common.h"

namespace {

// state[0] = step (incremented here), [1] = |g| before, [2] = |g| after, [3] = clip factor
template <typename real>
__global__ __launch_bounds__(1024) void adam_prep_kernel(const real* __restrict__ grad, int64_t n,
                                                         const real* __restrict__ sumsq_in,
                                                         real* __restrict__ state, real clip,
                                                         real gscale) {
  __shared__ real red[16];
  real sq = 0;
  if (sumsq_in == nullptr) {
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) sq += grad[i] * grad[i];
    sq = block_sum(sq, red);
  } else {
    sq = sumsq_in[0];
  }
  if (threadIdx.x == 0) {
    const real before = sqrt(sq) * gscale;           // norm of the scaled gradient
    real coef = 1;
    if (clip > real(0)) coef = tmin(clip / (before + real(1e-6)), real(1));
    state[0] += real(1);
    state[1] = before;
    state[2] = before * coef;
    state[3] = coef * gscale;                        // factor applied to the raw gradient
  }
}

template <typename real>
__global__ __launch_bounds__(256) void adam_apply_kernel(real* __restrict__ p,
                                                         const real* __restrict__ grad,
                                                         real* __restrict__ m, real* __restrict__ v,
                                                         int64_t n, const real* __restrict__ state,
                                                         real lr, real b1, real b2, real eps,
                                                         real wd) {
  const real step = state[0], coef = state[3];
  const real bc1 = real(1) - pow(b1, step), bc2s = sqrt(real(1) - pow(b2, step));
  const real step_size = lr / bc1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    real g = grad[i] * coef;
    const real w = p[i];
    if (wd != real(0)) g += wd * w;
    const real mi = b1 * m[i] + (real(1) - b1) * g;
    const real vi = b2 * v[i] + (real(1) - b2) * g * g;
    m[i] = mi;
    v[i] = vi;
    p[i] = w - step_size * mi / (sqrt(vi) / bc2s + eps);
  }
}

template <typename real>
int adam_flat(real* param, const real* grad, real* m, real* v, int64_t n, real* state,
              const real* sumsq_in, real lr, real b1, real b2, real eps, real wd, real clip,
              real gscale, hipStream_t st) {
  TCE_CHECK_ARG(param && grad && m && v && state && n > 0, "adam_flat: null buffer / bad size");
  hipLaunchKernelGGL(adam_prep_kernel<real>, dim3(1), dim3(1024), 0, st, grad, n, sumsq_in,
                     state, clip, gscale);
  TCE_LAUNCH_CHECK();
  const unsigned grid = (unsigned)tmin<int64_t>(ceil_div(n, 256), 2048);
  hipLaunchKernelGGL(adam_apply_kernel<real>, dim3(grid), dim3(256), 0, st, param, grad, m, v, n,
                     state, lr, b1, b2, eps, wd);
  TCE_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" {

int tce_adam_flat_f32(float* param, const float* grad, float* m, float* v, int64_t n,
                      float* state, const float* sumsq_in, float lr, float beta1, float beta2,
                      float eps, float weight_decay, float clip, float grad_scale,
                      void* stream) {
  return adam_flat<float>(param, grad, m, v, n, state, sumsq_in, lr, beta1, beta2, eps,
                          weight_decay, clip, grad_scale, (hipStream_t)stream);
}
int tce_adam_flat_f64(double* param, const double* grad, double* m, double* v, int64_t n,
                      double* state, const double* sumsq_in, double lr, double beta1,
                      double beta2, double eps, double weight_decay, double clip,
                      double grad_scale, void* stream) {
  return adam_flat<double>(param, grad, m, v, n, state, sumsq_in, lr, beta1, beta2, eps,
                           weight_decay, clip, grad_scale, (hipStream_t)stream);
}

}  // extern "C"
