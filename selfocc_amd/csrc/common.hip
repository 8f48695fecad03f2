// common.hip — ABI version + thread-local error string of libselfocc_hip.so
#include "so_device.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void so_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int selfocc_abi_version(void) { return SELFOCC_ABI_VERSION; }
extern "C" const char *selfocc_last_error(void) { return g_err; }
