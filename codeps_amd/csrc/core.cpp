// ABI version + thread-local error string of libcodeps_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/codeps_hip.h"

static thread_local char g_err[512] = "";

void codeps_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int codeps_version(void) { return 5; }
extern "C" const char* codeps_last_error(void) { return g_err; }
