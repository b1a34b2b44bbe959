#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kvcache.py tests/test_gpu_pipeline_golden.py tests/test_gpu_cli.py -x -q > $O/r04x_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/r04x_pytest.log
echo "--- 25 images"; timeout -k 10 300 python tools/bench_generate.py --images 25 2> $O/r04x_g25.err | cut -c1-1200
echo "--- 8 images (groups)"; timeout -k 10 300 python tools/bench_generate.py --images 8 2> $O/r04x_g8.err | cut -c1-1200
echo "--- 8 images (one batch)"; QARIG_NO_GROUPS=1 timeout -k 10 300 python tools/bench_generate.py --images 8 2> $O/r04x_g8b.err | cut -c1-1200
echo "--- 12 images (one batch)"; QARIG_NO_GROUPS=1 timeout -k 10 300 python tools/bench_generate.py --images 12 2> $O/r04x_g12b.err | cut -c1-1200
echo "--- 12 images (groups)"; timeout -k 10 300 python tools/bench_generate.py --images 12 2> $O/r04x_g12.err | cut -c1-1200
echo "--- 4 images"; timeout -k 10 300 python tools/bench_generate.py 2> $O/r04x_g4.err | cut -c1-1200
