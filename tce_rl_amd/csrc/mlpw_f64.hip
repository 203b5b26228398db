// Wide / double-precision fused critic epoch, float64 (see mlpw_impl.h):
// D_in -> 256 -> 256 -> 1 (D_in <= 24: the LDS holds W1 next to the panels) and
// D_in -> 128 -> 128 -> 1.
#include "mlpw_entry.h"

extern "C" int tce_mlpw_supported(int din, int hidden, int elem_size);

#define MLPW_DISPATCH(REAL)                                                          \
  if (hidden == 256)                                                                 \
    return mlpw_launch<REAL, 256, 6>(a, workspace, grad, stats, max_workgroups, w2,  \
                                     ad, st);                                        \
  return mlpw_launch<REAL, 128, 10>(a, workspace, grad, stats, max_workgroups, w2,   \
                                    ad, st);

MLPW_DEFINE(f64, double)
