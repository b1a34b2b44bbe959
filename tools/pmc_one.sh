#!/bin/bash
# One rocprofv3 counter pass over a python program, aggregated per (kernel, grid) -- ON THE GPU BOX:
#   tools/pmc_one.sh <tag> <name-substring> "<COUNTER ...>" <python script> [args]
set -e -o pipefail
TAG=$1; PAT=$2; CTRS=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/${TAG}_p -- python3 "$@" > $OUT/${TAG}_p.log 2>&1
python3 - "$OUT" "$TAG" "$PAT" <<'PY'
import collections, csv, glob, sys
out, tag, pat = sys.argv[1:4]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(f"{out}/{tag}_p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            key = (r["Kernel_Name"].split("(")[0][-44:], r["Grid_Size"])
            c = agg[key][r["Counter_Name"]]
            c[0] += 1
            c[1] += float(r["Counter_Value"])
for key, cs in sorted(agg.items()):
    print(key, "  ".join(f"{name}={tot / n:.1f}" for name, (n, tot) in sorted(cs.items())))
PY
rm -rf $OUT/${TAG}_p
