// error state + version of the C ABI (include/eoe_hip.h)
#include <stdarg.h>
#include <stdio.h>
#include "../../include/eoe_hip.h"

thread_local char g_eoe_err[512] = {0};

int eoe_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_eoe_err, sizeof(g_eoe_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int eoe_abi_version(void) { return EOE_ABI_VERSION; }
extern "C" const char* eoe_last_error(void) { return g_eoe_err; }
