// Thread-local error string + ABI version.
#include <stdarg.h>

#include "sf_common.h"

static thread_local char g_err[512] = "";

void sf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {
int sf_abi_version(void) { return SF_ABI_VERSION; }
const char* sf_last_error_string(void) { return g_err; }
}
