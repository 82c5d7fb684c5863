// Error plumbing + version of liblab4d_hip.so
#include <stdarg.h>

#include "common.hpp"

namespace lab4d {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace lab4d

extern "C" const char* lab4d_last_error(void) { return lab4d::g_err; }
extern "C" int lab4d_version(void) { return 1; }
extern "C" const char* lab4d_arch(void) { return "gfx950"; }
