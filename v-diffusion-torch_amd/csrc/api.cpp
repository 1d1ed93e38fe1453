// api.cpp -- version / error plumbing of the C ABI (include/vdiff_hip.h)
#include <stdarg.h>
#include <stdio.h>
#include "../../include/vdiff_hip.h"

static thread_local char g_err[512] = "";

void vd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int vd_version(void) { return VD_VERSION; }
extern "C" const char* vd_last_error(void) { return g_err; }
