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

extern "C" {

int nsg_version(void) { return NSG_VERSION; }

const char *nsg_last_error_string(void) { return g_err; }

}  // extern "C"
