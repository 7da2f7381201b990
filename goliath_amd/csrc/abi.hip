// abi.hip -- library identification and error reporting for libgoliath_hip.so.
#include <cstdarg>
#include <cstdio>

#include "gol_common.h"

static thread_local char g_err[512] = "";

void gol_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* gol_version(void) { return "goliath_hip 0.1.0 gfx950"; }
extern "C" const char* gol_last_error(void) { return g_err; }
