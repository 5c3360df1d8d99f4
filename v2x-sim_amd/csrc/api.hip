// ABI version + per-thread error text of libv2x_amd.so (see include/v2x_amd.h).
#include "common.h"

static thread_local char g_err[512] = "";

void v2x_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int v2x_abi_version(void) { return V2X_AMD_ABI_VERSION; }
extern "C" const char *v2x_last_error(void) { return g_err; }
