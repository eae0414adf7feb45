// Thread-local error string + ABI version for libroma_hip.so.
#include <cstdarg>
#include <cstdio>
#include "../../include/roma_hip.h"

namespace roma {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace roma

extern "C" int roma_abi_version(void) { return ROMA_ABI_VERSION; }
extern "C" const char* roma_last_error(void) { return roma::g_err; }
