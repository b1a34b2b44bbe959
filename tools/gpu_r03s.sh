#!/bin/bash
TAG=${1:-r03s}
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
step() { local t=$1 log=$2; shift 2; timeout -k 10 "$t" "$@" > "$log" 2> "${log%.*}.err"; local rc=$?; echo "[$(basename "$log")] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; return $rc; }
step 900 $O/${TAG}_pytest.log python -m pytest tests/test_gpu_bf16.py tests/test_gpu_fp8.py tests/test_gpu_core.py tests/test_gpu_grouped.py -m gpu -x -q; tail -4 $O/${TAG}_pytest.log
QARIG_GEMM_SHAPES=1 step 300 $O/${TAG}_bench_c5.json python bench.py --config c5 --steps 6 --warmup 2 --no-cpu-baseline; cut -c1-400 $O/${TAG}_bench_c5.json; grep "^lp" $O/${TAG}_bench_c5.err | head -12
step 300 $O/${TAG}_bench_c2.json python bench.py --steps 10 --warmup 3 --no-cpu-baseline; cut -c1-300 $O/${TAG}_bench_c2.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03s_bench_c2.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['frac'])
PY
