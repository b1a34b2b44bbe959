#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout -k 10 400 python bench.py > $O/r05b_bench_c2.json 2> $O/r05b_bench_c2.err; cut -c1-200 $O/r05b_bench_c2.json
bash tools/gpu_profiles.sh > $O/r05b_profiles.log 2>&1; tail -5 $O/r05b_profiles.log
