#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_switches.py -x -q 2>&1 | grep -v "^  \|Warning" | tail -30
