#!/bin/bash
TAG=${1:-r03q}
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
step() { local t=$1 log=$2; shift 2; timeout -k 10 "$t" "$@" > "$log" 2> "${log%.*}.err"; local rc=$?; echo "[$(basename "$log")] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; return $rc; }
step 900 $O/${TAG}_pytest.log python -m pytest tests/test_gpu_conv.py tests/test_gpu_pipeline_golden.py -m gpu -x -q; tail -6 $O/${TAG}_pytest.log
step 300 $O/${TAG}_generate_c3_batched_beams.json python tools/bench_generate.py --batch-beams; cat $O/${TAG}_generate_c3_batched_beams.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- python3 tools/bench_generate.py --batch-beams > $O/${TAG}_stats.log 2>&1
cp $(find $O/${TAG}_stats -name '*_kernel_stats.csv' | head -1) $O/${TAG}_generate_kernel_stats.csv; rm -rf $O/${TAG}_stats
grep -i "conv" $O/${TAG}_generate_kernel_stats.csv | cut -c1-60,150-400
