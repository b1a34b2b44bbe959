#!/bin/bash
TAG=${1:-r03o}
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
step() { local t=$1 log=$2; shift 2; timeout -k 10 "$t" "$@" > "$log" 2> "${log%.*}.err"; local rc=$?; echo "[$(basename "$log")] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; return $rc; }
step 600 $O/${TAG}_pytest.log python -m pytest tests/test_gpu_kvcache.py -m gpu -x -q; tail -6 $O/${TAG}_pytest.log
step 300 $O/${TAG}_generate_c3_sequential.json python tools/bench_generate.py; cat $O/${TAG}_generate_c3_sequential.json
step 300 $O/${TAG}_generate_c3_batched_beams.json python tools/bench_generate.py --batch-beams; cat $O/${TAG}_generate_c3_batched_beams.json
