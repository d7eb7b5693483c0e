// ABI version + thread-local error string for the C boundary.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void elvis_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int elvis_abi_version(void) { return ELVIS_ABI_VERSION; }
extern "C" const char* elvis_last_error(void) { return g_err; }
