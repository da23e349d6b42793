// Library-level helpers: version, thread-local error string.
#include <stdarg.h>
#include "common.hpp"

static thread_local char g_err[512] = "";

void imt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int imt_version(void) { return 100; }
extern "C" const char* imt_last_error(void) { return g_err; }
