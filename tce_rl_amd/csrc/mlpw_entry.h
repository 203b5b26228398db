// extern "C" entry of the wide / fp64 fused critic epoch for one arithmetic
// type (included by mlpw_f32.hip and mlpw_f64.hip: two translation units so
// that the big kernels compile in parallel).
#pragma once
#include "mlpw_impl.h"

extern "C" int tce_xchg_adam_f32(void* xchg, float* param, float* grad, float* m, float* v,
                                 int64_t n, float* state, float* norms_out, float step, float lr,
                                 float beta1, float beta2, float eps, float weight_decay,
                                 float clip, float grad_scale, void* stream);
extern "C" int tce_xchg_adam_f64(void* xchg, double* param, double* grad, double* m, double* v,
                                 int64_t n, double* state, double* norms_out, double step,
                                 double lr, double beta1, double beta2, double eps,
                                 double weight_decay, double clip, double grad_scale,
                                 void* stream);

extern "C" int tce_adam_once_f32(float* param, const float* grad, float* m, float* v, int64_t n,
                                 float* state, float* norms_out, float step, float lr,
                                 float beta1, float beta2, float eps, float weight_decay,
                                 float clip, float grad_scale, void* stream);
extern "C" int tce_adam_once_f64(double* param, const double* grad, double* m, double* v,
                                 int64_t n, double* state, double* norms_out, double step,
                                 double lr, double beta1, double beta2, double eps,
                                 double weight_decay, double clip, double grad_scale,
                                 void* stream);

#define MLPW_DEFINE(SFX, REAL)                                                      \
  static int mlpw_critic_impl_##SFX(                                                \
      const REAL* x, int64_t env_stride, int64_t row_stride, int T, int64_t R,      \
      int din, int hidden, const REAL* w1, const REAL* b1, const REAL* w2,          \
      const REAL* b2, const REAL* w3, const REAL* b3, int act, const REAL* returns, \
      const REAL* old_values, REAL clip, REAL* values, REAL* workspace,             \
      REAL* partials, REAL* grad, REAL* stats, int max_workgroups,                  \
      REAL* adam_param, REAL* adam_m, REAL* adam_v, REAL* adam_state, REAL lr,      \
      REAL beta1, REAL beta2, REAL eps, REAL weight_decay, REAL adam_step,          \
      REAL grad_scale, void* xchg, void* stream, const int64_t* row_index) {        \
    TCE_CHECK_ARG(x && w1 && b1 && w2 && b2 && w3 && b3 && workspace,               \
                  "mlpw_critic: null buffer");                                      \
    TCE_CHECK_ARG(R > 0 && T > 0 && din >= 1 && din <= 40,                          \
                  "mlpw_critic: R, T > 0 and 1 <= D_in <= 40");                     \
    TCE_CHECK_ARG(act >= 0 && act <= 3, "mlpw_critic: unknown activation");         \
    TCE_CHECK_ARG(tce_mlpw_supported(din, hidden, (int)sizeof(REAL)),               \
                  "mlpw_critic: unsupported (D_in, hidden, dtype) combination");    \
    TCE_CHECK_ARG(partials ? (returns && grad && stats) : (values != nullptr),      \
                  "mlpw_critic: backward needs returns / grad / stats, forward "    \
                  "needs values");                                                  \
    TCE_CHECK_ARG(!(partials && clip > 0 && !old_values),                           \
                  "mlpw_critic: clipped loss needs old_values");                    \
    WArgs<REAL> a;                                                                  \
    a.x = x; a.env_stride = env_stride; a.row_stride = row_stride; a.T = T;         \
    a.R = R; a.din = din; a.act = act;                                              \
    a.w1 = w1; a.b1 = b1; a.b2 = b2; a.w3 = w3; a.b3 = b3;                          \
    a.ret = returns; a.old_v = old_values; a.clip = clip; a.values = values;        \
    a.partials = partials; a.P = (int)mlpw_num_params(din, hidden);                 \
    a.row_index = row_index;                                                        \
    TCE_CHECK_ARG(!row_index || (partials && !values),                              \
                  "mlpw_critic: a row index goes with the backward pass only");     \
    TCE_CHECK_ARG(!xchg || (partials && adam_param),                                \
                  "mlpw_critic: an exchange needs the backward pass with the fused " \
                  "Adam step");                                                     \
    /* env shards: gradient only in the slab reduction, the exchange + Adam */      \
    /* follow as ONE small launch (csrc/mlp_shared.h) */                            \
    WAdam<REAL> ad = {xchg ? nullptr : adam_param, adam_m, adam_v, adam_state, lr,  \
                      beta1, beta2, eps, weight_decay, adam_step};                  \
    hipStream_t st = (hipStream_t)stream;                                           \
    const auto go_ = [&]() -> int { MLPW_DISPATCH(REAL) };                          \
    {                                                                               \
      const int rc_ = go_();                                                        \
      if (rc_) return rc_;                                                          \
    }                                                                               \
    if (xchg)                                                                       \
      return tce_xchg_adam_##SFX(xchg, adam_param, grad, adam_m, adam_v, a.P,       \
                                 adam_state, stats + 2, adam_step, lr, beta1,       \
                                 beta2, eps, weight_decay, REAL(0), grad_scale,     \
                                 stream);                                           \
    return 0;                                                                       \
  }                                                                                 \
  extern "C" int tce_mlpw_critic_##SFX(                                             \
      const REAL* x, int64_t env_stride, int64_t row_stride, int T, int64_t R,      \
      int din, int hidden, const REAL* w1, const REAL* b1, const REAL* w2,          \
      const REAL* b2, const REAL* w3, const REAL* b3, int act, const REAL* returns, \
      const REAL* old_values, REAL clip, REAL* values, REAL* workspace,             \
      REAL* partials, REAL* grad, REAL* stats, int max_workgroups,                  \
      REAL* adam_param, REAL* adam_m, REAL* adam_v, REAL* adam_state, REAL lr,      \
      REAL beta1, REAL beta2, REAL eps, REAL weight_decay, REAL adam_step,          \
      REAL grad_scale, void* xchg, void* stream) {                                  \
    return mlpw_critic_impl_##SFX(x, env_stride, row_stride, T, R, din, hidden, w1, \
                                  b1, w2, b2, w3, b3, act, returns, old_values,     \
                                  clip, values, workspace, partials, grad, stats,   \
                                  max_workgroups, adam_param, adam_m, adam_v,       \
                                  adam_state, lr, beta1, beta2, eps, weight_decay,  \
                                  adam_step, grad_scale, xchg, stream, nullptr);    \
  }                                                                                 \
  /* One critic epoch in minibatches: see tce_mlp_critic_minibatch_f32 */           \
  /* (csrc/mlp.hip) -- the same contract for the wide / float64 value nets. */      \
  extern "C" int tce_mlpw_critic_minibatch_##SFX(                                   \
      const REAL* x, int64_t env_stride, int64_t row_stride, int T, int64_t R,      \
      int din, int hidden, const REAL* w1, const REAL* b1, const REAL* w2,          \
      const REAL* b2, const REAL* w3, const REAL* b3, int act, const REAL* returns, \
      const REAL* old_values, REAL clip, const int64_t* row_index,                  \
      int num_minibatches, REAL* workspace, REAL* partials, REAL* grad,             \
      REAL* stats, int max_workgroups, REAL* adam_param, REAL* adam_m,              \
      REAL* adam_v, REAL* adam_state, REAL lr, REAL beta1, REAL beta2, REAL eps,    \
      REAL weight_decay, REAL adam_step, REAL grad_clip, REAL grad_scale,           \
      void* xchg, void* stream) {                                                   \
    TCE_CHECK_ARG(row_index && num_minibatches >= 1 && R >= num_minibatches,        \
                  "mlpw_critic_minibatch: row index missing / more minibatches "    \
                  "than rows");                                                     \
    TCE_CHECK_ARG(adam_param && adam_m && adam_v && adam_state && partials &&       \
                      grad && stats,                                                \
                  "mlpw_critic_minibatch: optimizer / gradient buffers missing");   \
    const int64_t P = mlpw_num_params(din, hidden);                                 \
    TCE_CHECK_ARG(!(grad_clip > 0) || P <= (1 << 17),                               \
                  "mlpw_critic_minibatch: clipping needs num_params <= 2^17");      \
    const bool fused = !(grad_clip > 0);                                            \
    const int64_t base = R / num_minibatches, extra = R % num_minibatches;          \
    int64_t off = 0;                                                                \
    for (int mb = 0; mb < num_minibatches; ++mb) {                                  \
      const int64_t len = base + (mb < extra ? 1 : 0);                              \
      REAL* st4 = stats + 4 * mb;                                                   \
      const REAL step = adam_step + (REAL)mb;                                       \
      int rc = mlpw_critic_impl_##SFX(                                              \
          x, env_stride, row_stride, T, len, din, hidden, w1, b1, w2, b2, w3, b3,   \
          act, returns, old_values, clip, nullptr, workspace, partials, grad, st4,  \
          max_workgroups, fused ? adam_param : nullptr, adam_m, adam_v,             \
          adam_state, lr, beta1, beta2, eps, weight_decay, step, grad_scale,        \
          fused ? xchg : nullptr, stream, row_index + off);                         \
      if (rc) return rc;                                                            \
      if (!fused) {                                                                 \
        rc = xchg ? tce_xchg_adam_##SFX(xchg, adam_param, grad, adam_m, adam_v, P,  \
                                        adam_state, st4 + 2, step, lr, beta1,       \
                                        beta2, eps, weight_decay, grad_clip,        \
                                        grad_scale, stream)                         \
                  : tce_adam_once_##SFX(adam_param, grad, adam_m, adam_v, P,        \
                                        adam_state, st4 + 2, step, lr, beta1,       \
                                        beta2, eps, weight_decay, grad_clip,        \
                                        grad_scale, stream);                        \
        if (rc) return rc;                                                          \
      }                                                                             \
      off += len;                                                                   \
    }                                                                               \
    return 0;                                                                       \
  }
