// Library-level entry points of libtce_hip.so (error string, version, device).
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

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

}  // extern "C"
