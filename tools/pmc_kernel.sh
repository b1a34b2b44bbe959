#!/bin/bash
# SQ counters of the kernels of one program, aggregated per (kernel, grid) -- run ON THE GPU BOX:
#   tools/pmc_kernel.sh <tag> <name-substring> <python script> [args]
# two counter passes (rocprofv3 --pmc with --kernel-trace only), prints per-launch averages.
set -e -o pipefail
TAG=$1; PAT=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_p1 -- python3 "$@" > $OUT/${TAG}_p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES --output-format csv -d $OUT/${TAG}_p2 -- python3 "$@" > $OUT/${TAG}_p2.log 2>&1
python3 - "$OUT" "$TAG" "$PAT" <<'PY'
import collections, csv, glob, sys
out, tag, pat = sys.argv[1:4]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for p in ("p1", "p2"):
    for f in glob.glob(f"{out}/{tag}_{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                key = (r["Kernel_Name"].split("(")[0][-44:], r["Grid_Size"])
                c = agg[key][r["Counter_Name"]]
                c[0] += 1
                c[1] += float(r["Counter_Value"])
for key, cs in sorted(agg.items()):
    print(key)
    for name, (n, tot) in sorted(cs.items()):
        print(f"    {name:30s} {tot / n:14.1f}   ({n} launches)")
PY
rm -rf $OUT/${TAG}_p1 $OUT/${TAG}_p2
