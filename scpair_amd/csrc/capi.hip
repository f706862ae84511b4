// C-ABI glue: error string, version.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"
#include "scpair_hip.h"

static thread_local char g_err[512] = "";

void scp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* scp_last_error(void) { return g_err; }
extern "C" int scp_version(void) { return 1; }
