// Library-level entry points of libtce_hip.so (error string, version, device).
#include "common.h"
#include <hip/hip_ext.h>
#include <string.h>

static thread_local char g_err[512] = "";

// an empty kernel of `tag` workgroups: a mark in a kernel trace (tce_marker)
__global__ void tce_marker_kernel() {}

// one wave busy for `ticks` of the 100 MHz wall clock (tce_spin_us)
__global__ void tce_spin_kernel(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

extern "C" {

void tce_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}

const char* tce_last_error(void) { return g_err; }

int tce_version(void) { return 1; }

// Number of visible HIP devices (0 when there is none); never throws.
int tce_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// A stream whose kernels run only on `cus_per_xcd` compute units of every XCD,
// starting at unit `first_cu` (0..31) of the XCD: lets two independent kernel
// chains (critic epochs, policy epochs) own disjoint parts of the chip.  CU
// mask bit b addresses XCD b % 8, unit b / 8 (measured on MI355X:
// scripts/ubench_cumask.hip).  Returns the hipStream_t in *stream.
int tce_stream_create_cu_range(int first_cu, int cus_per_xcd, void** stream) {
  TCE_CHECK_ARG(stream && first_cu >= 0 && cus_per_xcd >= 1 && first_cu + cus_per_xcd <= 32,
                "stream_create_cu_range: bad CU range (32 units per XCD)");
  uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int j = first_cu; j < first_cu + cus_per_xcd; ++j)
    for (int x = 0; x < 8; ++x) {
      const int b = 8 * j + x;
      mask[b >> 5] |= 1u << (b & 31);
    }
  hipStream_t st = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, mask);
  if (e != hipSuccess) {
    tce_set_error(hipGetErrorString(e));
    return 2;
  }
  *stream = (void*)st;
  return 0;
}

// Put a mark into the stream that a kernel trace shows: an empty launch of `tag`
// workgroups named tce_marker_kernel (scripts/rocpd_stats.py --between-markers
// A B keeps the dispatches between the marks with A and B workgroups: the timed
// window of bench.py without its warm-up and its roofline launches).
int tce_marker(int tag, void* stream) {
  TCE_CHECK_ARG(tag >= 1 && tag <= 65535, "marker: 1 <= tag <= 65535");
  hipLaunchKernelGGL(tce_marker_kernel, dim3((unsigned)tag), dim3(64), 0, (hipStream_t)stream);
  TCE_LAUNCH_CHECK();
  return 0;
}

int tce_spin_us(double us, void* stream) {
  TCE_CHECK_ARG(us > 0 && us <= 100000.0, "spin_us: 0 < us <= 100000");
  hipLaunchKernelGGL(tce_spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                     (unsigned long long)(us * 100.0));
  TCE_LAUNCH_CHECK();
  return 0;
}

int tce_stream_destroy(void* stream) {
  if (stream && hipStreamDestroy((hipStream_t)stream) != hipSuccess) return 2;
  return 0;
}

}  // extern "C"
