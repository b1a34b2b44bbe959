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

QarigOptions g_qarig_opt;

// Sets a kernel-selection option (see QarigOptions); returns its previous value, or INT_MIN for an
// unknown name (the error string says which names exist).  Not synchronised: callers set options
// between launches (tests, tools), not concurrently with them.
extern "C" int qarig_set_option(const char* name, int value) {
    struct Entry { const char* name; int* slot; };
    const Entry table[] = {
        {"gemm_dma", &g_qarig_opt.gemm_dma},       {"gemm_pair", &g_qarig_opt.gemm_pair},
        {"bmu_cs", &g_qarig_opt.bmu_cs},           {"bmu_groups", &g_qarig_opt.bmu_groups},
        {"bmu_coarse", &g_qarig_opt.bmu_coarse},   {"attn_qw", &g_qarig_opt.attn_qw},
        {"attn_bw", &g_qarig_opt.attn_bw},         {"lp_big", &g_qarig_opt.lp_big},
        {"lp_mfma16", &g_qarig_opt.lp_mfma16},     {"convt_pair", &g_qarig_opt.convt_pair},
        {"conv_ring", &g_qarig_opt.conv_ring},     {"gemm_xcd_splits", &g_qarig_opt.gemm_xcd_splits},
        {"decode_stream", &g_qarig_opt.decode_stream}, {"decode_rows", &g_qarig_opt.decode_rows},
        {"gemm_tile64", &g_qarig_opt.gemm_tile64}, {"gemm_x3", &g_qarig_opt.gemm_x3},
    };
    if (name)
        for (const Entry& e : table)
            if (strcmp(e.name, name) == 0) {
                const int old = *e.slot;
                *e.slot = value;
                return old;
            }
    qarig_set_error("set_option: unknown option %s (gemm_dma, gemm_pair, bmu_cs, bmu_groups, bmu_coarse, attn_qw, "
                    "attn_bw, lp_big, lp_mfma16, convt_pair, conv_ring, gemm_xcd_splits, decode_stream, decode_rows, gemm_tile64, gemm_x3)", name ? name : "(null)");
    return -2147483647 - 1;
}
