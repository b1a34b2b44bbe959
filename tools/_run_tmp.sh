#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for n in 8 4; do echo "--- $n images"; timeout -k 10 300 python tools/bench_generate.py --images $n 2> $O/r04x_g$n.err | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['accepted_tokens_per_s'], [(s['seconds'], s['host_enqueue_seconds']) for s in j['stages']])"; done
