// cbim_api.cpp — version / backend / thread-local error string of the C ABI (include/cbim_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/cbim_hip.h"

static thread_local char g_err[512] = "";

void cbim_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int cbim_version(void) { return 100; }
extern "C" const char* cbim_backend(void) {
#ifdef CBIM_EMU
  return "emu";
#else
  return "hip-gfx950";
#endif
}
extern "C" const char* cbim_last_error_string(void) { return g_err; }
