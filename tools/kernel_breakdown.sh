#!/bin/bash
# Per-kernel time of one bench configuration (rocprofv3 --kernel-trace --stats), run ON THE GPU BOX:
#   tools/kernel_breakdown.sh <tag> <steps-profiled> [bench args]
# writes gpurun_out/<tag>_kernel_stats.csv and prints ms/step per kernel.
set -e -o pipefail
TAG=$1; STEPS=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
W=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- python3 bench.py --steps $STEPS --warmup $W --no-kernel-events --no-cpu-baseline --no-side-configs "$@" > gpurun_out/${TAG}_stats.log 2>&1
F=$(find gpurun_out/${TAG}_stats -name '*kernel_stats.csv' | head -1)
cp "$F" gpurun_out/${TAG}_kernel_stats.csv
rm -rf gpurun_out/${TAG}_stats
python3 - "$TAG" "$STEPS" "$W" <<'PY'
import csv, sys
tag, steps, w = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
n = steps + w
rows = list(csv.DictReader(open(f"gpurun_out/{tag}_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:26]:
    print(r["Name"][:96].ljust(96), f'{int(r["Calls"]) / n:8.1f}/step {float(r["TotalDurationNs"]) / n / 1e6:8.3f} ms/step {r["Percentage"]:>6}%')
print("kernel time per step (ms):", round(tot / n / 1e6, 3), "(includes the bmu side measurement and warm-up effects)")
PY
