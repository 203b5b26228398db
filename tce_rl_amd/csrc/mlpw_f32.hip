// Wide fused critic epoch, float32 (see mlpw_impl.h): D_in -> 256 -> 256 -> 1.
#include "mlpw_entry.h"

extern "C" int tce_mlpw_supported(int din, int hidden, int elem_size);

#define MLPW_DISPATCH(REAL)                                                          \
  return mlpw_launch<REAL, 256, 10>(a, workspace, grad, stats, max_workgroups, w2,   \
                                    ad, st);

MLPW_DEFINE(f32, float)

extern "C" {

// (D_in, hidden width, element size) combinations the kernels are built for
int tce_mlpw_supported(int din, int hidden, int elem_size) {
  if (din < 1 || din > 40) return 0;
  if (elem_size == 4) return hidden == 256;
  if (elem_size == 8) return hidden == 128 || (hidden == 256 && din <= 24);
  return 0;
}

int tce_mlpw_grid(void) { return mlpw_cu_count(); }

int64_t tce_mlpw_num_params(int din, int hidden) { return mlpw_num_params(din, hidden); }

// workspace elements: W2 images + (backward) H1, dY2, dY1 [R][hidden]
int64_t tce_mlpw_workspace_len(int64_t R, int hidden, int backward) {
  return 2 * (int64_t)hidden * hidden + (backward ? 3 * mlpw_ws_rows(R) * hidden : 0);
}

}  // extern "C"
