// Error reporting + library identity for libdiagan_hip.so.
#include "common.h"
#include <string.h>

namespace diagan {
static thread_local char g_err[512] = "";
char* err_buf() { return g_err; }
int set_err(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace diagan

DIAGAN_API const char* diagan_last_error(void) { return diagan::err_buf(); }
DIAGAN_API int diagan_abi_version(void) { return 1; }
DIAGAN_API const char* diagan_target_arch(void) { return "gfx950"; }
