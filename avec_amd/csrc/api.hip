// Error plumbing + version for libavec_hip.so
#include "common.h"
#include "avec_hip.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void avec_set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char* avec_last_error() { return g_err; }
extern "C" int avec_version() { return AVEC_ABI_VERSION; }
