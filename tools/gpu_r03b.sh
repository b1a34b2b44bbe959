#!/bin/bash
TAG=${1:-r03b}
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
step() { local t=$1 log=$2; shift 2; timeout -k 10 "$t" "$@" > "$log" 2> "${log%.*}.err"; local rc=$?; echo "[$(basename "$log")] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; return $rc; }
step 300 $O/${TAG}_dispatch_c4.log python tools/dispatch_trace.py --config c4; tail -80 $O/${TAG}_dispatch_c4.log
step 300 $O/${TAG}_bench_c4_graph.json python bench.py --config c4 --steps 10 --warmup 3 --graph; cat $O/${TAG}_bench_c4_graph.json | cut -c1-400
step 300 $O/${TAG}_bench_c2_graph.json python bench.py --steps 10 --warmup 3 --graph --no-cpu-baseline; cat $O/${TAG}_bench_c2_graph.json | cut -c1-400
QARIG_MLP_GROUPED=1 step 300 $O/${TAG}_bench_c2_grouped.json python bench.py --steps 10 --warmup 3 --no-cpu-baseline; cat $O/${TAG}_bench_c2_grouped.json | cut -c1-1500
step 300 $O/${TAG}_bench_c2.json python bench.py --steps 10 --warmup 3 --no-cpu-baseline; cat $O/${TAG}_bench_c2.json | cut -c1-1500
step 300 $O/${TAG}_host_c4.log python tools/host_profile.py --config c4; tail -40 $O/${TAG}_host_c4.log
