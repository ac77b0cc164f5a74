// C-ABI plumbing: error reporting and version for libsegmentron_hip.so.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

namespace seg {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return 2;
  }
  return 0;
}
}  // namespace seg

extern "C" const char* seg_last_error(void) { return seg::g_err; }
extern "C" int seg_version(void) { return 1; }
