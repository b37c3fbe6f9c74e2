// Error reporting and version for libnsg.so.
#include "nsg_common.h"
#include <string.h>

namespace {
thread_local char g_err[512] = "";
}

void nsg_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int nsg_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int nsg_check_launch(const char *what)
{
    const hipError_t err = hipGetLastError();
    if (err == hipSuccess) return NSG_OK;
    nsg_set_error("%s: %s", what, hipGetErrorString(err));
    return (int)err;
}

int nsg_lds_opt_in(LdsOptIn &once, std::initializer_list<const void *> kernels, size_t lds_bytes, const char *what)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return nsg_fail((int)e, "%s: hipGetDevice: %s", what, hipGetErrorString(e));
    const uint64_t bit = 1ull << (dev & 63);
    if (once.devices.load(std::memory_order_acquire) & bit) return NSG_OK;
    for (const void *k : kernels) {
        e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return nsg_fail((int)e, "%s: cannot reserve %zu bytes of LDS on device %d", what, lds_bytes, dev);
    }
    once.devices.fetch_or(bit, std::memory_order_release);
    return NSG_OK;
}

extern "C" {

int nsg_version(void) { return NSG_VERSION; }

const char *nsg_last_error_string(void) { return g_err; }

}  // extern "C"
