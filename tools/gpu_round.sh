#!/bin/bash
# One GPU call: tests, benches, micro-benchmarks; everything lands in gpurun_out/<tag>_*.
# A step that times out or is killed (rc 124 / 137) ends the call: no further GPU step after it.
TAG=${1:-r02b}
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
step 900 $O/${TAG}_pytest.log python -m pytest tests -m gpu -x -q; tail -4 $O/${TAG}_pytest.log
if [ "${SKIP_BENCH:-0}" = "1" ]; then exit 0; fi
step 300 $O/${TAG}_attn_bench.log python tools/attn_bench.py; cat $O/${TAG}_attn_bench.log
QARIG_CPU_BASELINE_SECONDS=3 step 300 $O/${TAG}_bench_c2.json python bench.py --steps 10 --warmup 3; cat $O/${TAG}_bench_c2.json
step 300 $O/${TAG}_bench_c4.json python bench.py --config c4 --steps 10 --warmup 3; cat $O/${TAG}_bench_c4.json
step 400 $O/${TAG}_bench_c5.json python bench.py --config c5 --steps 3 --warmup 1; cat $O/${TAG}_bench_c5.json; tail -3 $O/${TAG}_bench_c5.err
step 400 $O/${TAG}_bench_c5_fp8.json python bench.py --config c5 --precision fp8 --steps 3 --warmup 1; cat $O/${TAG}_bench_c5_fp8.json | cut -c1-300
step 200 $O/${TAG}_bmu_bench.log python tools/bmu_bench.py; cat $O/${TAG}_bmu_bench.log
step 200 $O/${TAG}_smoke.log python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"; tail -2 $O/${TAG}_smoke.log
