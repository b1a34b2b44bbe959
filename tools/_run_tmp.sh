#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_core.py tests/test_gpu_codebook.py -x -q -k "bmu or codebook or Codebook" > $O/r04z_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/r04z_pytest.log
BMU_PHASES=1 timeout -k 10 300 python tools/bmu_bench.py > $O/r04z_bmu.log 2>&1; cat $O/r04z_bmu.log
echo "--- plain tensor (no image)"; BMU_FROZEN=0 timeout -k 10 300 python tools/bmu_bench.py 2>&1 | head -3
echo "--- coarse forced"; QARIG_BMU_COARSE=1 timeout -k 10 300 python tools/bmu_bench.py 2>&1 | head -6
echo "--- coarse forced, plain"; BMU_FROZEN=0 QARIG_BMU_COARSE=1 timeout -k 10 300 python tools/bmu_bench.py 2>&1 | head -6
