#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
QARIG_GEMM_X3=1 timeout -k 10 300 python bench.py --no-side-configs --no-cpu-baseline > $O/r04x_bench_c2_x3.json 2> $O/r04x_bench_c2_x3.err; cut -c1-400 $O/r04x_bench_c2_x3.json
timeout -k 10 300 python bench.py --no-side-configs --no-cpu-baseline > $O/r04x_bench_c2_f32.json 2> $O/r04x_bench_c2_f32.err; cut -c1-300 $O/r04x_bench_c2_f32.json
QARIG_GEMM_X3=1 timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/r04x_pytest_x3.log 2>&1; echo "pytest rc=$?"; tail -6 $O/r04x_pytest_x3.log
