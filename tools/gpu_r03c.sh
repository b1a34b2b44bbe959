#!/bin/bash
TAG=${1:-r03c}
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
step() { local t=$1 log=$2; shift 2; timeout -k 10 "$t" "$@" > "$log" 2> "${log%.*}.err"; local rc=$?; echo "[$(basename "$log")] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; return $rc; }
step 300 $O/${TAG}_dispatch_c4.log python tools/dispatch_trace.py --config c4; grep -A12 "aten ops that launch" $O/${TAG}_dispatch_c4.log | cut -c1-230
step 600 $O/${TAG}_pytest_dp.log python -m pytest tests/test_gpu_dp.py tests/test_gpu_bf16.py -m gpu -x -q; tail -15 $O/${TAG}_pytest_dp.log
