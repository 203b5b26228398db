// extern "C" entry of the wide / fp64 fused critic epoch for one arithmetic
// type (included by mlpw_f32.hip and mlpw_f64.hip: two translation units so
// that the big kernels compile in parallel).
#pragma once
#include "mlpw_impl.h"

#define MLPW_DEFINE(SFX, REAL)                                                      \
  extern "C" int tce_mlpw_critic_##SFX(                                             \
      const REAL* x, int64_t env_stride, int64_t row_stride, int T, int64_t R,      \
      int din, int hidden, const REAL* w1, const REAL* b1, const REAL* w2,          \
      const REAL* b2, const REAL* w3, const REAL* b3, int act, const REAL* returns, \
      const REAL* old_values, REAL clip, REAL* values, REAL* workspace,             \
      REAL* partials, REAL* grad, REAL* stats, int max_workgroups,                  \
      REAL* adam_param, REAL* adam_m, REAL* adam_v, REAL* adam_state, REAL lr,      \
      REAL beta1, REAL beta2, REAL eps, REAL weight_decay, REAL adam_step,          \
      void* stream) {                                                               \
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
    WAdam<REAL> ad = {adam_param, adam_m, adam_v, adam_state, lr, beta1, beta2,     \
                      eps, weight_decay, adam_step};                                \
    hipStream_t st = (hipStream_t)stream;                                           \
    MLPW_DISPATCH(REAL)                                                             \
    return 0;                                                                       \
  }
