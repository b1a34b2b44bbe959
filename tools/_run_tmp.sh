#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_switches.py tests/test_gpu_grouped.py -x -q 2>&1 | tail -3
QARIG_GEMM_X3=1 timeout -k 10 600 python -m pytest tests/test_gpu_grouped.py tests/test_gpu_transformer.py tests/test_gpu_pipeline_golden.py -x -q 2>&1 | tail -2
QARIG_GEMM_X3=1 timeout -k 10 300 python bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline > $O/r04x_bench_c4_x3.json 2> $O/r04x_bench_c4_x3.err; python -c "
import json; j=json.loads(open('$O/r04x_bench_c4_x3.json').read().strip().splitlines()[-1]); print('c4 x3', j['ms_per_step'], j['value'], j['roofline']['achieved'], j['roofline']['frac'])"
QARIG_GEMM_X3=1 timeout -k 10 300 python bench.py --no-side-configs --no-cpu-baseline > $O/r04x_bench_c2_x3.json 2> $O/r04x_bench_c2_x3.err; python -c "
import json; j=json.loads(open('$O/r04x_bench_c2_x3.json').read().strip().splitlines()[-1]); print('c2 x3', j['ms_per_step'], j['value'], j['roofline']['achieved'], j['roofline']['frac'])"
