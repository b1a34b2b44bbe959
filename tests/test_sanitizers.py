"""Host side of the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY section 5:
'host ASan/UBSan on the C-ABI shim').  No GPU: every entry point of include/qarig.h is called
with (a) NULL / zero arguments, (b) plausible non-NULL pointers with degenerate, negative and
huge extents, (c) well-formed arguments -- validation then passes, the workspace / grid / LDS
arithmetic runs, and the launch itself fails with 'no ROCm-capable device' (status -2), which
is as far as a CPU box can go.  The run must finish without a sanitizer report and every call
must return a status (never crash); device pointers are never dereferenced on the host."""
import os
import subprocess
import sys

import pytest

from conftest import PKG, ROOT

DRIVER = r'''
import ctypes, itertools, sys
sys.path.insert(0, sys.argv[2])
from qarig import _lib
lib = ctypes.CDLL(sys.argv[1])
P, I, L, Z, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t, ctypes.c_float
calls = 0
statuses = {}
# arguments that are HOST arrays of device pointers (read by the host side): given a real 16-entry table
HOST_TABLES = {"qarig_gemm_f32_grouped": (1, 4, 7, 12, 13, 15, 18, 24)}
def table(p):
    return (P * 16)(*[p] * 16)
for name, (res, args) in sorted(_lib.SIGNATURES.items()):
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = res, args
    if name in ("qarig_last_error", "qarig_target_arch", "qarig_version", "qarig_set_option"):
        continue      # (qarig_set_option returns a previous VALUE, not a status: exercised by hand below)
    ints = [i for i, a in enumerate(args) if a in (I, L, Z)]
    def build(ptr, ival, fval):
        out = []
        for i, a in enumerate(args):
            if a is P and i in HOST_TABLES.get(name, ()): out.append(table(ptr) if ptr is not None else None)
            elif a is P: out.append(ptr)
            elif a is F: out.append(fval)
            elif a is ctypes.c_char_p: out.append(None)
            else: out.append(ival)
        return out
    variants = [build(None, 0, 0.0), build(0x10000, 0, 1.0), build(0x10000, -1, 1.0),
                build(0x10000, 1, 1.0), build(0x10000, 8, 2.8284271), build(0x10000, 64, 2.8284271),
                build(0x10000, 128, 8.0), build(0x10000, 256, 8.0), build(0x10000, 512, 8.0),
                build(0x10007, 128, 8.0),                                     # misaligned pointers
                build(0x10000, 2**31 - 1 if any(a is I for a in args) else 2**40, 8.0)]
    # one integer at a time pushed to an extreme, the others plausible
    for i in ints:
        for extreme in (-7, 0, 3, 2**31 - 1 if args[i] is I else 2**40):
            v = build(0x10000, 128, 8.0)
            v[i] = extreme
            variants.append(v)
    for v in variants:
        r = fn(*v)
        calls += 1
        if res is I:
            statuses.setdefault(name, set()).add(int(r))
# well-formed README-shaped calls of the hot entry points: validation passes, the dispatch /
# workspace / grid arithmetic runs, the launch fails for want of a device
X = 0x7f0000000000
X2 = 0x7f1000000000          # a second fake device address (entries that refuse aliased in/out)
def call(name, *a):
    global calls
    r = getattr(lib, name)(*a)
    calls += 1
    assert r in (-2, -3), (name, r)          # launch failed / workspace refused: never 0, never a crash
for (M, N, K, ak, bk, sk) in ((16384, 2048, 512, 1, 1, 1), (16384, 512, 2048, 1, 0, 1), (2048, 512, 16384, 0, 0, 8),
                              (640, 2048, 16384, 0, 0, 6), (100, 513, 96, 1, 1, 1), (64, 2048, 512, 1, 1, 1)):
    call("qarig_gemm_f32", X, K if ak else M, ak, X, K if bk else N, bk, X, N, M, N, K, X, None, 0, X, N, 1,
         None, 0, 0, sk, 0, None, X, 1 << 40, None)
call("qarig_gemm_lp", X, 512, X, 512, 0, X, 2048, 16384, 2048, 512, X, None, 0, None, 0, 1, None, 0, 0, 0, 1, 0,
     X, 2048, None, 0, None, 0, None)
call("qarig_gemm_lp", X, 2048, X, 512, 1, X, 512, 2048, 512, 16384, None, None, 0, None, 0, 0, None, 0, 0, 0, 8, 0,
     None, 0, None, 0, X, 1 << 40, None)
call("qarig_gemm_f8", X, 512, X, 512, X, X, X, 2048, 16384, 2048, 512, X, None, 0, None, 0, 1, X, 2048, None, 0, None)
call("qarig_cast_fp8", X, 16384 * 512, X, X, X, X, None)
for (M, N, K, ak, bk, sk) in ((2048, 2048, 512, 1, 1, 1), (2048, 512, 2048, 0, 0, 4)):      # the paired kernel's launches
    call("qarig_gemm_f32", X, K if ak else M, ak, X, K if bk else N, bk, X, N, M, N, K, None, None, 0, None, 0, 0,
         None, 0, 0, sk, 0, None, X, 1 << 40, None)
# grouped launches: q/k/v forward, second layer with a split reduction, the summed input gradient of 14 blocks,
# weight gradients with row sums
for (G, M, N, K, ak, bk, sk, sumg, rs) in ((3, 2048, 2048, 512, 1, 1, 1, 0, 0), (3, 2048, 512, 2048, 1, 1, 4, 0, 0),
                                           (14, 2048, 512, 2048, 1, 0, 1, 1, 0), (3, 512, 2048, 2048, 0, 0, 4, 0, 1),
                                           (16, 2048, 2048, 512, 1, 0, 1, 0, 0)):
    call("qarig_gemm_f32_grouped", G, table(X), K if ak else M, ak, table(X), K if bk else N, bk, table(X), N, M, N, K,
         table(X) if ak and bk else None, None, 0, None, 0, 0, None, 0, 0, sk, rs, sumg, table(X) if rs else None,
         X, 1 << 40, None)
for (N, C, H, W, p, K) in ((64, 4, 32, 32, 2, 512), (64, 4, 32, 32, 32, 512), (64, 4, 64, 64, 1, 8192), (3, 4, 12, 20, 2, 77)):
    call("qarig_bmu_fwd", X, N, C, H, W, p, p, X, K, C * p * p, X, X, 1 << 40, None)
    call("qarig_bmu_fwd_prepared", X, N, C, H, W, p, p, X, K, C * p * p, X, X, 1 << 40, X, None)
for (N, Sq, Sk, H, d, causal) in ((64, 256, 256, 64, 8, 1), (2, 4096, 4096, 64, 8, 1), (2, 4096, 1024, 64, 8, 0),
                                  (3, 12, 5, 2, 16, 0), (1, 130, 130, 2, 64, 0)):
    call("qarig_attention_fwd", X, X, X, N, Sq, Sk, H, d, causal, float(d) ** 0.5, X, X, None)
    call("qarig_attention_bwd", X, X, X, X, X, X, N, Sq, Sk, H, d, causal, float(d) ** 0.5, X, X, X, X, None)
for (N, Cin, H, W, Cout, k, st) in ((4, 3, 128, 128, 256, 3, 1), (4, 256, 128, 128, 512, 3, 2), (4, 512, 32, 32, 4, 3, 1)):
    call("qarig_conv2d_fwd", X, N, Cin, H, W, X, X, Cout, k, st, 1, 1, X, X, None)
    call("qarig_conv_wgrad", X, N, Cout, H // st, W // st, X, Cin, H, W, k, st, 1, X, X, 1 << 40, None)
call("qarig_conv_transpose2d_fwd", X, 4, 512, 32, 32, X, X, 256, 1, X, X, X, 1 << 40, 0, None)
# the ring kernels' launch geometry: 3x3 forward with a scratch buffer, its input gradient, ConvTranspose at a
# batch of 16, the band form of the Gaussian neighbourhood
for (N, Cin, H, W, Cout) in ((16, 256, 128, 128, 256), (16, 512, 32, 32, 512), (1, 48, 16, 24, 256), (2, 130, 12, 10, 140)):
    call("qarig_conv2d_fwd_ws", X, N, Cin, H, W, X, X, Cout, 3, 1, 1, 1, X, X, X, 1 << 40, 0, None)
    call("qarig_conv2d_bwd_data", X, N, Cout, H, W, X, Cin, 3, 1, 1, H, W, X, X, 1 << 40, None)
    call("qarig_conv_wgrad", X, N, Cout, H, W, X, Cin, H, W, 3, 1, 1, X, X, 1 << 40, None)
call("qarig_conv_transpose2d_fwd", X, 16, 256, 64, 64, X, X, 256, 1, X, X, X, 1 << 40, 0, None)
# few images: split reductions (slabs behind the packed weights), the re-ordering launch skipped
call("qarig_conv2d_fwd_ws", X, 4, 512, 32, 32, X, X, 512, 3, 1, 1, 1, X, X, X, 1 << 40, 1, None)
call("qarig_conv2d_fwd_ws", X, 4, 256, 128, 128, X, X, 3, 3, 1, 1, 2, X, None, X, 1 << 40, 0, None)
assert lib.qarig_conv2d_fwd_workspace_bytes_n(4, 512, 32, 32, 512, 3, 1) == 512 * 512 * 9 * 4 + 4 * 4 * 512 * 1024 * 4
assert lib.qarig_conv2d_fwd_workspace_bytes_n(16, 512, 32, 32, 512, 3, 1) == 512 * 512 * 9 * 4
assert lib.qarig_conv_transpose2d_workspace_bytes_n(4, 512, 32, 32, 256) == 16 * 512 * 256 * 4 + 4 * 4 * 256 * 4096 * 4
assert lib.qarig_conv2d_bwd_data_workspace_bytes_n(4, 512, 16, 16, 512, 3, 1) > 512 * 512 * 9 * 4
assert lib.qarig_conv2d_bwd_data_workspace_bytes_n(4, 512, 32, 32, 512, 3, 2) == 512 * 512 * 9 * 4
assert lib.qarig_conv2d_fwd_workspace_bytes_n(4, 256, 64, 64, 512, 3, 2) > 256 * 512 * 9 * 4
assert lib.qarig_conv_transpose2d_bwd_data_workspace_bytes_n(4, 512, 16, 16, 256) > 16 * 512 * 256 * 4
call("qarig_conv2d_bwd_data", X, 4, 512, 16, 16, X, 512, 3, 1, 1, 16, 16, X, X, 1 << 40, None)
call("qarig_conv2d_fwd_ws", X, 4, 256, 64, 64, X, X, 512, 3, 2, 1, 1, X, None, X, 1 << 40, 0, None)
call("qarig_conv_transpose2d_bwd_data_ws", X, 4, 256, 16, 16, X, 512, X, X, 1 << 40, None)
# the decode step's fused Linear launches: AdaLN form on 3 stacked weights, affine form, gate multiply
call("qarig_gemm_skinny_ln_f32", X, 512, 1e-5, None, None, X, X, 512, X, 512, 2048 * 512, X, 2048, 4 * 2048, X, 2048,
     None, 0, 3, 4, 2048, 512, 1, None)
call("qarig_gemm_skinny_ln_f32", X, 512, 1e-5, X, X, None, None, 0, X, 512, 0, X, 2048, 0, X, 0, None, 0, 1, 64, 2048,
     512, 1, None)
call("qarig_gemm_skinny_ln_f32", X, 2048, 0.0, None, None, None, None, 0, X, 2048, 0, X, 512, 0, X, 0, X, 512, 1, 16,
     512, 2048, 1, None)
# the weight-streaming decode kernels at README shapes: q/k/v first layer with the AdaLN row, second layer of
# the feed-forward MLP with the gate row, residual layer, ragged classifier; one step's other launches
call("qarig_decode_linear_f32", X, 512, 0, 1e-5, None, None, X, X, 0, X, 512, 2048 * 512, X, 2048, None, 0, None, 0,
     X, 2048, 4 * 2048, 3, 4, 2048, 512, 1, None)
call("qarig_decode_linear_f32", X, 2048, 0, 0.0, None, None, None, None, 0, X, 2048, 0, X, 0, None, 0, X, 0,
     X, 512, 0, 1, 16, 512, 2048, 1, None)
call("qarig_decode_linear_f32", X, 512, 0, 0.0, None, None, None, None, 0, X, 512, 0, X, 0, X, 512, None, 0,
     X, 512, 0, 1, 4, 512, 512, 1, None)
call("qarig_decode_linear_f32", X, 2048, 0, 0.0, None, None, None, None, 0, X, 2048, 0, X, 0, None, 0, None, 0,
     X, 513, 0, 1, 16, 513, 2048, 0, None)
call("qarig_decode_embed", X, 16, 512, 1025, X, X, X, 0, 256, X, 63 * 512, X, X, X, None)
call("qarig_decode_attention", X, X, X, X, X, 16, 64, 8, 0, X, 256, 256 * 512, 256 * 8, 8, 8.0 ** 0.5, X, 0, X, None)
call("qarig_decode_attention", X, None, None, X, X, 4, 64, 8, 64, None, 64, 64 * 512, 8, 512, 8.0 ** 0.5, None, 0, X, None)
call("qarig_decode_sample", X, 513, 16, 513, 1.0, 512, 1, 512, X, None, X, 3, 4, 1024, 1, 4, X, X, X, None, None)
call("qarig_decode_decide", X, 4, 4, 4, 16, X, X, X, X, X, None)
call("qarig_decode_rows", X, X, X, X, 14, 4, 4, 64, 3, 8, 256, 0, None)
call("qarig_decode_rows", X, X, X, None, 14, 4, 1, 64, 3, 8, 256, 1, None)
call("qarig_decode_commit", X, 4, 4, 4, X, X, 260, X, None)
call("qarig_decode_advance", X, 4, None)
call("qarig_som_band", X, 8192, 4, 1779.0, 222, X2, None)
call("qarig_som_band", X, 512, 16, 0.43, 4, X2, None)
so = lib.qarig_set_option
so.restype, so.argtypes = I, [ctypes.c_char_p, I]
assert so(None, 1) == -2**31 and so(b"no_such_option", 1) == -2**31 and so(b"", 0) == -2**31
assert so(b"gemm_pair", 1) == -1 and so(b"gemm_pair", -1) == 1
assert so(b"x" * 4000, 7) == -2**31           # a long unknown name must not overrun the error string
calls += 6
buf = ctypes.create_string_buffer(8)          # a too-short buffer must be respected
lib.qarig_last_error.argtypes = [ctypes.c_char_p, Z]
lib.qarig_last_error(buf, 8)
assert buf.raw[7:8] == b"\x00"
bad = {n: s for n, s in statuses.items() if not s <= {0, -1, -2, -3, 1}}
assert not bad, bad
# a GPU-less box never reports success for a launch
launched_ok = [n for n, s in statuses.items() if 0 in s and not n.endswith("_supported")]
print("CALLS", calls, "ENTRY_POINTS", len(statuses), "OK_STATUS", launched_ok)
'''


def test_host_entry_points_under_asan_ubsan(tmp_path):
    import build as qbuild
    so = qbuild.build_sanitizer_lib()
    rt = qbuild.asan_runtime()
    if not os.path.exists(rt):
        pytest.skip("clang ASan runtime not found")
    script = tmp_path / "driver.py"
    script.write_text(DRIVER)
    env = dict(os.environ, LD_PRELOAD=rt,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=66",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=67")
    r = subprocess.run([sys.executable, str(script), so, PKG], capture_output=True, text=True, timeout=600,
                       env=env, cwd=ROOT)
    report = r.stdout[-3000:] + r.stderr[-6000:]
    assert "ERROR: AddressSanitizer" not in report and "runtime error:" not in report, report
    assert r.returncode == 0, report
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("CALLS")][0]
    assert int(line.split()[1]) > 1300 and int(line.split()[3]) >= 38, line
