#!/bin/bash
TAG=${1:-r03n}
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
step() { local t=$1 log=$2; shift 2; timeout -k 10 "$t" "$@" > "$log" 2> "${log%.*}.err"; local rc=$?; echo "[$(basename "$log")] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; return $rc; }
step 600 $O/${TAG}_pytest.log python -m pytest tests/test_gpu_transformer.py tests/test_gpu_kvcache.py -m gpu -x -q; tail -6 $O/${TAG}_pytest.log
step 300 $O/${TAG}_bench_c5.json python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline; cut -c1-1500 $O/${TAG}_bench_c5.json
timeout -k 10 900 bash tools/profile_round.sh r03_c5 --config c5 > $O/${TAG}_profile_c5.log 2>&1; tail -3 $O/${TAG}_profile_c5.log
