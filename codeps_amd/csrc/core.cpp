// ABI version + thread-local error string of libcodeps_hip.so (and, built with -DCODEPS_DIAG_LIB, of libcodeps_diag.so).
#include <stdarg.h>
#include <stdio.h>

#include <mutex>
#include <set>
#include <utility>

#include "common.h"

static thread_local char g_err[512] = "";

void codeps_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#ifdef CODEPS_DIAG_LIB
extern "C" int codeps_diag_version(void) { return 1; }
extern "C" const char* codeps_diag_last_error(void) { return g_err; }
#else
extern "C" int codeps_version(void) { return 11; }
extern "C" const char* codeps_last_error(void) { return g_err; }
#endif

// ---- per-(kernel, device) latch of the dynamic-LDS limit
static std::mutex g_attr_mu;
static std::set<std::pair<const void*, int>> g_attr_done;
static thread_local bool g_attr_failed = false;

void codeps_allow_lds(const void* kernel, int bytes, const char* name) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    // fast path (hundreds of conv launches per eager step): this thread registered this (kernel, device) last time it asked
    static thread_local const void* last_kernel = nullptr;
    static thread_local int last_dev = -1;
    if (kernel == last_kernel && dev == last_dev) return;
    std::lock_guard<std::mutex> lock(g_attr_mu);
    if (g_attr_done.count({kernel, dev})) { last_kernel = kernel; last_dev = dev; return; }
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        codeps_set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) failed on device %d: %s", name, bytes, dev, hipGetErrorString(e));
        g_attr_failed = true;
        return;
    }
    g_attr_done.insert({kernel, dev});
    last_kernel = kernel; last_dev = dev;
}

bool codeps_take_attr_failure(void) {
    const bool f = g_attr_failed;
    g_attr_failed = false;
    return f;
}
