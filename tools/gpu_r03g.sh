#!/bin/bash
TAG=${1:-r03g}
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
step() { local t=$1 log=$2; shift 2; timeout -k 10 "$t" "$@" > "$log" 2> "${log%.*}.err"; local rc=$?; echo "[$(basename "$log")] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; return $rc; }
step 600 $O/${TAG}_pytest_bmu.log python -m pytest tests/test_gpu_core.py -m gpu -x -q -k "bmu"; tail -12 $O/${TAG}_pytest_bmu.log
step 200 $O/${TAG}_bmu_bench.log python tools/bmu_bench.py; cat $O/${TAG}_bmu_bench.log
QARIG_BMU_COARSE=0 step 200 $O/${TAG}_bmu_bench_exact.log python tools/bmu_bench.py; cat $O/${TAG}_bmu_bench_exact.log
step 600 $O/${TAG}_pytest_misc.log python -m pytest tests/test_gpu_dp.py tests/test_gpu_kvcache.py -m gpu -x -q; tail -8 $O/${TAG}_pytest_misc.log
step 300 $O/${TAG}_bench_c4.json python bench.py --config c4 --steps 10 --warmup 3 --no-kernel-events; cut -c1-900 $O/${TAG}_bench_c4.json
step 300 $O/${TAG}_bench_c2.json python bench.py --steps 10 --warmup 3 --no-cpu-baseline; cut -c1-900 $O/${TAG}_bench_c2.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_probe -- python3 tools/copy_probe.py > $O/${TAG}_probe.log 2>&1
python3 tools/copy_probe_report.py $O/${TAG}_probe $O/${TAG}_probe.log > $O/${TAG}_probe_report.txt 2>&1; cat $O/${TAG}_probe_report.txt; rm -rf $O/${TAG}_probe
