#!/bin/bash
# round 3, first GPU call: grouped GEMM parity + micro-benchmark + config-4 shard A/B + ATen attribution
TAG=${1:-r03a}
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
step() {   # step <seconds> <logfile> <cmd...>
    local t=$1 log=$2; shift 2
    timeout -k 10 "$t" "$@" > "$log" 2> "${log%.*}.err"
    local rc=$?
    echo "[$(basename "$log")] rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed at its limit: stopping"; exit $rc; fi
    return $rc
}
step 600 $O/${TAG}_pytest_new.log python -m pytest tests/test_gpu_grouped.py tests/test_gpu_core.py -m gpu -x -q || { tail -30 $O/${TAG}_pytest_new.log; exit 1; }
tail -3 $O/${TAG}_pytest_new.log
step 300 $O/${TAG}_grouped_bench.log python tools/gemm_grouped_bench.py; cat $O/${TAG}_grouped_bench.log
step 300 $O/${TAG}_bench_c4.json python bench.py --config c4 --steps 10 --warmup 3; cat $O/${TAG}_bench_c4.json
QARIG_MLP_GROUPED=0 step 300 $O/${TAG}_bench_c4_nogroup.json python bench.py --config c4 --steps 10 --warmup 3; cat $O/${TAG}_bench_c4_nogroup.json
QARIG_CROSS_KV_GROUPING=layer step 300 $O/${TAG}_bench_c4_layer.json python bench.py --config c4 --steps 10 --warmup 3; cat $O/${TAG}_bench_c4_layer.json
step 300 $O/${TAG}_aten_c4.log python tools/aten_trace.py --config c4; cat $O/${TAG}_aten_c4.log
step 900 $O/${TAG}_pytest.log python -m pytest tests -m gpu -x -q; tail -5 $O/${TAG}_pytest.log
