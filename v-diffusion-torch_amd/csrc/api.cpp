// api.cpp -- version / error plumbing of the C ABI (include/vdiff_hip.h)
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <hip/hip_runtime_api.h>
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

// persistent / round-sized launches size themselves by the CU count of the device they run on: cached per device id (a process
// may drive several devices), lock-free (any thread may launch)
int vd_cu_count(void) {
    constexpr int MAXDEV = 64;
    static std::atomic<int> cache[MAXDEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return 256;
    int n = cache[dev].load(std::memory_order_relaxed);
    if (n > 0) return n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cache[dev].store(n, std::memory_order_relaxed);
    return n;
}

// CUs the persistent convolution launches leave free (vd_set_reserved_cus; include/vdiff_hip.h)
static std::atomic<int> g_reserved{-1};           // -1: not set yet -> VD_RESERVE_CUS or 0
extern "C" int vd_reserved_cus(void) {
    int n = g_reserved.load(std::memory_order_relaxed);
    if (n >= 0) return n;
    const char* e = getenv("VD_RESERVE_CUS");
    n = e ? atoi(e) : 0;
    if (n < 0) n = 0;
    g_reserved.store(n, std::memory_order_relaxed);
    return n;
}
extern "C" int vd_set_reserved_cus(int32_t n) {
    if (n < 0 || n > vd_cu_count() - 8) { vd_set_error("vd_set_reserved_cus: %d outside [0, CUs - 8]", n); return -1; }
    const int prev = vd_reserved_cus();
    g_reserved.store(n, std::memory_order_relaxed);
    return prev;
}
int vd_persistent_cus(void) {
    const int ncu = vd_cu_count(), r = vd_reserved_cus();
    return r <= ncu - 8 ? ncu - r : 8;
}
