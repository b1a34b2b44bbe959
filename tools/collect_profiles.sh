#!/bin/bash
# Copies the outputs of tools/gpu_validate.sh <tag> (gpurun_out/<tag>_*) into profiles/r04_* (run here, after the
# gpurun call has merged them back):  tools/collect_profiles.sh r04y
TAG=${1:?tag}
cd "$(dirname "$0")/.."
for f in bench_c2_gemm_x3.json bench_c4_gemm_x3.json gemm_x3_sweep.log bench_c2.json bench_c3.json bench_c4.json bench_c5.json generate_c3_sequential.json generate_c3_one_by_one.json generate_c3_first_call.json generate_c3_25_images.json \
         generate_c3_batched_beams.json decode_step_rows4.json decode_step_rows16.json decode_chain_rows4.json \
         decode_chain_rows16.json bmu_bench.log c1_autoencoder.log; do
    [ -s gpurun_out/${TAG}_$f ] && cp gpurun_out/${TAG}_$f profiles/r04_$f
done
tail -4 gpurun_out/${TAG}_pytest.log | grep -E "passed|failed" > profiles/r04_pytest_gpu_summary.log
cat profiles/r04_pytest_gpu_summary.log
