// Library-wide C-ABI plumbing: version, last-error string.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "qarig_common.h"

static thread_local char g_err[512] = "";

extern "C" void qarig_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int qarig_version(void) { return 100; }  // 0.1.0

// Copies the calling thread's last error message into buf (NUL-terminated).
extern "C" int qarig_last_error(char* buf, size_t n) {
    if (!buf || n == 0) return QARIG_ERR_ARG;
    strncpy(buf, g_err, n - 1);
    buf[n - 1] = '\0';
    return QARIG_OK;
}

extern "C" const char* qarig_target_arch(void) { return "gfx950"; }
