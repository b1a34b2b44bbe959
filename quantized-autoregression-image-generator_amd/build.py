"""Builds lib/libqarig_hip.so from csrc/*.hip with hipcc for gfx950 (in-tree).

No torch, no pybind: the library is a plain C-ABI shared object (include/qarig.h)
loaded with ctypes by qarig/_lib.py.  Objects are rebuilt only when their source
(or a header) is newer.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(HERE, "lib", "libqarig_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function"]
# per-file additions.  decode.hip: the decode step's launches are bound by dependent memory round trips;
# with the first 16 dwords of the kernel arguments preloaded into SGPRs by the command processor their first
# loads do not wait for a scalar-cache miss on the argument segment (the kernels list what they need first).
FILE_FLAGS = {"decode.hip": ["-mllvm", "-amdgpu-kernarg-preload-count=16"]}
if os.environ.get("QARIG_NO_KERNARG_PRELOAD") == "1":       # ablation builds (tools/)
    FILE_FLAGS = {}


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build_lib(verbose=True, force=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdr_time = max(os.path.getmtime(h) for h in hdrs) if hdrs else 0
    jobs = []
    objs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s[:-4] + ".o")
        objs.append(obj)
        if force or _newer(src, obj) or (os.path.exists(obj) and hdr_time > os.path.getmtime(obj)):
            jobs.append([HIPCC, *FLAGS, *FILE_FLAGS.get(s, []), "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


SAN_LIB = os.path.join(HERE, "lib", "libqarig_hip_san.so")
SAN_FLAGS = ["--offload-arch=gfx950", "-O1", "-g", "-fPIC", "-std=c++17", "-ffp-contract=off",
             "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-gpu-sanitize",
             "-fno-omit-frame-pointer", "-Wno-unused-function"]


def build_sanitizer_lib(verbose=False):
    """Test-only build: the HOST side of every csrc/*.hip (argument validation, workspace and
    launch-geometry arithmetic, dispatch) under AddressSanitizer + UndefinedBehaviorSanitizer;
    device code is compiled normally (GPU ASan / xnack builds are not available on this pool).
    tests/test_sanitizers.py loads it in a subprocess with the ASan runtime preloaded."""
    obj_dir = os.path.join(HERE, "lib", "obj_san")
    os.makedirs(obj_dir, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".h"))
    jobs, objs = [], []
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(obj_dir, s[:-4] + ".o")
        objs.append(obj)
        if _newer(src, obj) or hdr_time > os.path.getmtime(obj):
            jobs.append([HIPCC, *SAN_FLAGS, "-c", src, "-o", obj])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc (sanitizer build) failed:\n" + r.stdout + r.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(SAN_LIB):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address,undefined",
             "-fno-gpu-sanitize", "-o", SAN_LIB, *objs])
    return SAN_LIB


def asan_runtime():
    r = subprocess.run([os.path.join(os.path.dirname(HIPCC), "..", "lib", "llvm", "bin", "clang"),
                        "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    return r.stdout.strip()


if __name__ == "__main__":
    if "--sanitizers" in sys.argv:
        print(build_sanitizer_lib(verbose=True))
    else:
        print(build_lib(force="--force" in sys.argv))
