// Wide fused critic epoch, float32 (see mlpw_impl.h): D_in -> 256 -> 256 -> 1.
#include "mlpw_entry.h"

extern "C" int tce_mlpw_supported(int din, int hidden, int elem_size);

// D_in <= 24 (box pushing, table tennis: 22): 6 features per lane group
// instead of 10 -- 24 instead of 40 k-steps in layer 1 and in dW1, a W1 image
// of 28 instead of 44 columns: 3.20 -> 3.12 ms per C3 epoch.
// (scripts/mlpw_variant.py builds one shape with -DMLPW_F32_KPG=...)
#ifdef MLPW_TRY_H128
// (experiment: the 128-wide critic of C2 on the two-launch kernels, scripts/time_mlpw128.py)
#define MLPW_DISPATCH(REAL)                                                          \
  if (hidden == 128)                                                                 \
    return mlpw_launch<REAL, 128, 10>(a, workspace, grad, stats, max_workgroups, w2, \
                                      ad, st);                                       \
  return mlpw_launch<REAL, 256, 6>(a, workspace, grad, stats, max_workgroups, w2,    \
                                   ad, st);
#elif defined(MLPW_F32_KPG)
#define MLPW_DISPATCH(REAL)                                                          \
  return mlpw_launch<REAL, 256, MLPW_F32_KPG>(a, workspace, grad, stats,             \
                                              max_workgroups, w2, ad, st);
#else
#define MLPW_DISPATCH(REAL)                                                          \
  if (din <= 24)                                                                     \
    return mlpw_launch<REAL, 256, 6>(a, workspace, grad, stats, max_workgroups, w2,  \
                                     ad, st);                                        \
  return mlpw_launch<REAL, 256, 10>(a, workspace, grad, stats, max_workgroups, w2,   \
                                    ad, st);
#endif

MLPW_DEFINE(f32, float)

extern "C" {

// (D_in, hidden width, element size) combinations the kernels are built for
int tce_mlpw_supported(int din, int hidden, int elem_size) {
  if (din < 1 || din > 40) return 0;
#ifdef MLPW_TRY_H128
  if (elem_size == 4) return hidden == 256 || hidden == 128;
#endif
  if (elem_size == 4) return hidden == 256;
  if (elem_size == 8) return hidden == 128 || (hidden == 256 && din <= 24);
  return 0;
}

int tce_mlpw_grid(void) { return mlpw_cu_count() * WCfg<float>::WGS; }

int64_t tce_mlpw_num_params(int din, int hidden) { return mlpw_num_params(din, hidden); }

// workspace elements: W2 images + (backward) H1, dY2, dY1 [R][hidden]
int64_t tce_mlpw_workspace_len(int64_t R, int hidden, int backward) {
  return 2 * (int64_t)hidden * hidden + (backward ? 3 * mlpw_ws_rows(R) * hidden : 0);
}

}  // extern "C"
