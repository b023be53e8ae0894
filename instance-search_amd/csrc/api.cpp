// api.cpp -- error reporting and version of libisx (host only).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/isx.h"

static thread_local char g_err[512] = "";

void isx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" __attribute__((visibility("default"))) const char* isx_last_error(void) { return g_err; }
extern "C" __attribute__((visibility("default"))) int isx_version(void) { return 110; }
