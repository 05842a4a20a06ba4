// api.cpp — error plumbing of the C ABI (host only).
#include <stdarg.h>
#include <stdio.h>

#include "czc_common.h"

namespace czc {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  set_error("HIP error %d (%s) at %s:%d in `%s`", (int)e, hipGetErrorString(e), file, line, what);
  return CZC_ERR_HIP;
}
}  // namespace czc

extern "C" const char* czc_last_error(void) { return czc::g_err; }
extern "C" int czc_abi_version(void) { return CZC_ABI_VERSION; }
