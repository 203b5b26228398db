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
  real step_size, bc2s;
  adam_coef(lr, b1, b2, step, step_size, bc2s);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    real w = p[i], mi = m[i], vi = v[i];
    adam_elem(grad[i] * coef, w, mi, vi, b1, b2, eps, wd, step_size, bc2s);
    m[i] = mi;
    v[i] = vi;
    p[i] = w;
  }
}

// The same step as ONE launch for small buffers (the sharded update calls it
// behind every gradient all-reduce, 100 times per iteration: there the two
// launches of adam_flat + a copy of the norms were 3 of the 5 launches of a
// critic epoch).  Every workgroup forms |g|^2 over the WHOLE buffer in the same
// order (n <= 2^17 elements from L2: cheaper than a second launch), then
// applies its own slice; the step count comes from the host (`step` = count
// including this update) and is stored to state[0].  norms_out (nullable)
// receives {|g| before, |g| after clipping}.
constexpr int ADAM1_BT = 1024, ADAM1_MAX_BLOCKS = 32;
template <typename real>
__global__ __launch_bounds__(ADAM1_BT) void adam_once_kernel(
    real* __restrict__ p, const real* __restrict__ grad, real* __restrict__ m,
    real* __restrict__ v, int64_t n, real* __restrict__ state, real* __restrict__ norms_out,
    real step, real lr, real b1, real b2, real eps, real wd, real clip, real gscale) {
  __shared__ real red[16];
  real sq = 0;
  for (int64_t i = threadIdx.x; i < n; i += ADAM1_BT) sq += grad[i] * grad[i];
  sq = block_sum(sq, red);
  const real before = sqrt(sq) * gscale;
  real coef = 1;
  if (clip > real(0)) coef = tmin(clip / (before + real(1e-6)), real(1));
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    state[0] = step;
    state[1] = before;
    state[2] = before * coef;
    state[3] = coef * gscale;
    if (norms_out) { norms_out[0] = before; norms_out[1] = before * coef; }
  }
  const real cg = coef * gscale;
  real step_size, bc2s;
  adam_coef(lr, b1, b2, step, step_size, bc2s);
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t i0 = blockIdx.x * per, i1 = tmin<int64_t>(n, i0 + per);
  for (int64_t i = i0 + threadIdx.x; i < i1; i += ADAM1_BT) {
    real w = p[i], mi = m[i], vi = v[i];
    adam_elem(grad[i] * cg, w, mi, vi, b1, b2, eps, wd, step_size, bc2s);
    m[i] = mi;
    v[i] = vi;
    p[i] = w;
  }
}

template <typename real>
int adam_once(real* param, const real* grad, real* m, real* v, int64_t n, real* state,
              real* norms_out, real step, real lr, real b1, real b2, real eps, real wd, real clip,
              real gscale, hipStream_t st) {
  TCE_CHECK_ARG(param && grad && m && v && state && n > 0 && n <= (1 << 17) && step >= real(1),
                "adam_once: null buffer / n outside [1, 2^17] / step < 1");
  const unsigned grid = (unsigned)tmin<int64_t>(ceil_div(n, 4 * ADAM1_BT), ADAM1_MAX_BLOCKS);
  hipLaunchKernelGGL(adam_once_kernel<real>, dim3(grid), dim3(ADAM1_BT), 0, st, param, grad, m, v,
                     n, state, norms_out, step, lr, b1, b2, eps, wd, clip, gscale);
  TCE_LAUNCH_CHECK();
  return 0;
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

int tce_adam_once_f32(float* param, const float* grad, float* m, float* v, int64_t n,
                      float* state, float* norms_out, float step, float lr, float beta1,
                      float beta2, float eps, float weight_decay, float clip, float grad_scale,
                      void* stream) {
  return adam_once<float>(param, grad, m, v, n, state, norms_out, step, lr, beta1, beta2, eps,
                          weight_decay, clip, grad_scale, (hipStream_t)stream);
}
int tce_adam_once_f64(double* param, const double* grad, double* m, double* v, int64_t n,
                      double* state, double* norms_out, double step, double lr, double beta1,
                      double beta2, double eps, double weight_decay, double clip,
                      double grad_scale, void* stream) {
  return adam_once<double>(param, grad, m, v, n, state, norms_out, step, lr, beta1, beta2, eps,
                           weight_decay, clip, grad_scale, (hipStream_t)stream);
}

}  // extern "C"
