#!/bin/bash
TAG=${1:-r03p}
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
step() { local t=$1 log=$2; shift 2; timeout -k 10 "$t" "$@" > "$log" 2> "${log%.*}.err"; local rc=$?; echo "[$(basename "$log")] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; return $rc; }
step 300 $O/${TAG}_decode_step_rows4.json python tools/decode_step_probe.py --rows 4; cat $O/${TAG}_decode_step_rows4.json
step 300 $O/${TAG}_decode_step_rows16.json python tools/decode_step_probe.py --rows 16; cat $O/${TAG}_decode_step_rows16.json
step 300 $O/${TAG}_decode_step_base4.json python tools/decode_step_probe.py --rows 4 --base; cat $O/${TAG}_decode_step_base4.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- python3 tools/decode_step_probe.py --rows 4 > $O/${TAG}_stats.log 2>&1
cp $(find $O/${TAG}_stats -name '*_kernel_stats.csv' | head -1) $O/${TAG}_decode_step_kernel_stats.csv; rm -rf $O/${TAG}_stats
head -20 $O/${TAG}_decode_step_kernel_stats.csv | cut -c1-150
