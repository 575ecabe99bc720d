/* Status / error plumbing of libvfe. */
#include "vfe_common.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void vfe_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* vfe_last_error(void) { return g_err; }

extern "C" const char* vfe_status_string(int s) {
    switch (s) {
        case VFE_OK: return "ok";
        case VFE_E_INVALID: return "invalid argument";
        case VFE_E_HIP: return "HIP runtime error";
        case VFE_E_NOMEM: return "out of memory / workspace too small";
        case VFE_E_CAPACITY: return "buffer capacity exceeded";
        case VFE_E_NODEVICE: return "no usable gfx950 device";
        default: return "unknown status";
    }
}

extern "C" const char* vfe_version(void) { return "vfe 0.1 (gfx950)"; }

extern "C" int vfe_device_count(int* count) {
    VFE_REQUIRE(count, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        vfe_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return VFE_E_NODEVICE;
    }
    *count = n;
    return VFE_OK;
}
