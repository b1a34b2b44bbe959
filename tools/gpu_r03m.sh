#!/bin/bash
TAG=${1:-r03m}
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
step() { local t=$1 log=$2; shift 2; timeout -k 10 "$t" "$@" > "$log" 2> "${log%.*}.err"; local rc=$?; echo "[$(basename "$log")] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; return $rc; }
step 1000 $O/${TAG}_pytest.log python -m pytest tests -m gpu -x -q; tail -6 $O/${TAG}_pytest.log
step 300 $O/${TAG}_generate_c3_sequential.json python tools/bench_generate.py; cat $O/${TAG}_generate_c3_sequential.json
step 300 $O/${TAG}_generate_c3_batched_beams.json python tools/bench_generate.py --batch-beams; cat $O/${TAG}_generate_c3_batched_beams.json
step 300 $O/${TAG}_generate_c3_full_window.json python tools/bench_generate.py --no-kv-cache; cat $O/${TAG}_generate_c3_full_window.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_gen_stats -- python3 tools/bench_generate.py --batch-beams > $O/${TAG}_gen_stats.log 2>&1
cp $(find $O/${TAG}_gen_stats -name '*_kernel_stats.csv' | head -1) $O/${TAG}_generate_c3_kernel_stats.csv; rm -rf $O/${TAG}_gen_stats
QARIG_CPU_BASELINE_SECONDS=3 step 300 $O/${TAG}_bench_c2.json python bench.py --steps 10 --warmup 3; cut -c1-1500 $O/${TAG}_bench_c2.json
timeout -k 10 900 bash tools/profile_round.sh ${TAG}_c2 > $O/${TAG}_profile_c2.log 2>&1; tail -3 $O/${TAG}_profile_c2.log
