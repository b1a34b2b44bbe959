#!/bin/bash
# rocprofv3 evidence of the final build: kernel stats + FETCH/WRITE + SQ counters for c2, c4, c5; decode-step kernel stats
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout -k 10 500 bash tools/profile_round.sh r03_c2 > $O/r03v_profile_c2.log 2>&1; tail -2 $O/r03v_profile_c2.log
timeout -k 10 500 bash tools/profile_round.sh r03_c4 --config c4 > $O/r03v_profile_c4.log 2>&1; tail -2 $O/r03v_profile_c4.log
timeout -k 10 500 bash tools/profile_round.sh r03_c5 --config c5 > $O/r03v_profile_c5.log 2>&1; tail -2 $O/r03v_profile_c5.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03v_gen -- python3 tools/bench_generate.py --batch-beams > $O/r03v_gen.log 2>&1
cp $(find $O/r03v_gen -name '*_kernel_stats.csv' | head -1) $O/r03_generate_c3_kernel_stats.csv; rm -rf $O/r03v_gen
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03v_dec -- python3 tools/decode_step_probe.py --rows 4 > $O/r03v_dec.log 2>&1
cp $(find $O/r03v_dec -name '*_kernel_stats.csv' | head -1) $O/r03_decode_step_kernel_stats.csv; rm -rf $O/r03v_dec
ls $O | grep "r03_" | head -40
