#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_kvcache.py tests/test_gpu_pipeline_golden.py tests/test_gpu_cli.py -x -q > $O/r04x_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/r04x_pytest.log
timeout -k 10 300 python tools/bench_generate.py > $O/r04x_gen_seq.json 2> $O/r04x_gen_seq.err && cat $O/r04x_gen_seq.json
timeout -k 10 300 python tools/bench_generate.py --rebuild-models > $O/r04x_gen_first.json 2> $O/r04x_gen_first.err && cat $O/r04x_gen_first.json
timeout -k 10 300 python tools/bench_generate.py --batch-beams > $O/r04x_gen_bb.json 2> $O/r04x_gen_bb.err && cat $O/r04x_gen_bb.json
QARIG_GEN_TIMING=1 timeout -k 10 300 python tools/bench_generate.py > /dev/null 2> $O/r04x_gen_timing.err; grep "qarig generate" $O/r04x_gen_timing.err | tail -8
