// Host-side runtime glue of liboct_hip.so: error reporting, version, device probing.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "../../include/oct_hip.h"

static thread_local char g_err[512] = "";

void oct_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int oct_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    oct_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return OCT_E_LAUNCH;
  }
  return OCT_OK;
}

extern "C" {
const char* oct_version_string(void) { return "oct_hip 0.2.2 (gfx950)"; }
int oct_version(void) { return OCT_VERSION; }
int oct_get_last_error(char* buf, size_t len) {
  if (!buf || len == 0) return OCT_E_INVALID;
  strncpy(buf, g_err, len - 1);
  buf[len - 1] = 0;
  return OCT_OK;
}
int oct_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}
}
