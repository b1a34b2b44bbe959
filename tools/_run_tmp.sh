#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_cli.py -x -q > $O/r04x_pytest.log 2>&1; echo "pytest rc=$?"; tail -30 $O/r04x_pytest.log
