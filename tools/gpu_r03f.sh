#!/bin/bash
TAG=${1:-r03f}
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
step() { local t=$1 log=$2; shift 2; timeout -k 10 "$t" "$@" > "$log" 2> "${log%.*}.err"; local rc=$?; echo "[$(basename "$log")] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; return $rc; }
step 900 $O/${TAG}_pytest.log python -m pytest tests -m gpu -x -q; tail -5 $O/${TAG}_pytest.log
step 300 $O/${TAG}_bench_c4.json python bench.py --config c4 --steps 10 --warmup 3; cut -c1-1700 $O/${TAG}_bench_c4.json
ROWS=2048 SPLITS=0,1,2,4,8 step 300 $O/${TAG}_gemm_splits_2048.log python tools/gemm_bench.py; cat $O/${TAG}_gemm_splits_2048.log
timeout -k 10 900 bash tools/profile_round.sh ${TAG}_c4 --config c4 --eager > $O/${TAG}_profile_c4.log 2>&1; tail -3 $O/${TAG}_profile_c4.log
