#!/bin/bash
# Full validation of a build: every GPU test, the entry-point smoke, the four bench configurations, the
# generation / decode-step / BMU micro-benchmarks (everything lands in gpurun_out/<tag>_*).
TAG=${1:-val}
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
step() { local t=$1 log=$2; shift 2; timeout -k 10 "$t" "$@" > "$log" 2> "${log%.*}.err"; local rc=$?; echo "[$(basename "$log")] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; return $rc; }
step 1100 $O/${TAG}_pytest.log python -m pytest tests -m gpu -x -q; tail -4 $O/${TAG}_pytest.log
step 200 $O/${TAG}_smoke.log python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"; tail -2 $O/${TAG}_smoke.log
step 400 $O/${TAG}_bench_c2.json python bench.py; cut -c1-330 $O/${TAG}_bench_c2.json
step 300 $O/${TAG}_bench_c4.json python bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline; cut -c1-330 $O/${TAG}_bench_c4.json
step 400 $O/${TAG}_bench_c5.json python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline; cut -c1-330 $O/${TAG}_bench_c5.json
step 300 $O/${TAG}_bench_c2_gemm_x3.json python bench.py --gemm-x3 --no-side-configs --no-cpu-baseline; cut -c1-330 $O/${TAG}_bench_c2_gemm_x3.json
step 300 $O/${TAG}_bench_c4_gemm_x3.json python bench.py --config c4 --gemm-x3 --steps 10 --warmup 3 --no-cpu-baseline; cut -c1-330 $O/${TAG}_bench_c4_gemm_x3.json
step 300 $O/${TAG}_gemm_x3_sweep.log env SWEEP=gemm_x3=0,1 python tools/gemm_bench.py; tail -10 $O/${TAG}_gemm_x3_sweep.log
step 400 $O/${TAG}_bench_c3.json python bench.py --config c3 --steps 1 --warmup 1; cut -c1-330 $O/${TAG}_bench_c3.json
step 300 $O/${TAG}_generate_c3_sequential.json python tools/bench_generate.py; cat $O/${TAG}_generate_c3_sequential.json
step 300 $O/${TAG}_generate_c3_one_by_one.json python tools/bench_generate.py --one-by-one; cat $O/${TAG}_generate_c3_one_by_one.json
step 300 $O/${TAG}_generate_c3_first_call.json python tools/bench_generate.py --rebuild-models; cat $O/${TAG}_generate_c3_first_call.json
step 300 $O/${TAG}_generate_c3_25_images.json python tools/bench_generate.py --images 25; cat $O/${TAG}_generate_c3_25_images.json
step 300 $O/${TAG}_generate_c3_batched_beams.json python tools/bench_generate.py --batch-beams; cat $O/${TAG}_generate_c3_batched_beams.json
step 300 $O/${TAG}_decode_step_rows4.json python tools/decode_step_probe.py --rows 4; cat $O/${TAG}_decode_step_rows4.json
step 300 $O/${TAG}_decode_step_rows16.json python tools/decode_step_probe.py --rows 16; cat $O/${TAG}_decode_step_rows16.json
step 200 $O/${TAG}_decode_chain_rows4.json python tools/decode_chain_bench.py --rows 4; cat $O/${TAG}_decode_chain_rows4.json
step 200 $O/${TAG}_decode_chain_rows16.json python tools/decode_chain_bench.py --rows 16; cat $O/${TAG}_decode_chain_rows16.json
step 200 $O/${TAG}_bmu_bench.log python tools/bmu_bench.py; tail -12 $O/${TAG}_bmu_bench.log
step 200 $O/${TAG}_c1_autoencoder.log python tools/c1_autoencoder.py; tail -2 $O/${TAG}_c1_autoencoder.log
