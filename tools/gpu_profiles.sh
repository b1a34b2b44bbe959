#!/bin/bash
# rocprofv3 evidence of the final build: kernel stats + FETCH/WRITE + SQ counters for c2; kernel stats for c4, c5;
# decode-step and generation kernel stats (everything lands in gpurun_out/r04_*; profiles/summarize.py and a copy
# bring the summaries into profiles/)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout -k 10 500 bash tools/profile_round.sh r04_c2 > $O/r04v_profile_c2.log 2>&1; tail -2 $O/r04v_profile_c2.log
timeout -k 10 300 bash tools/kernel_breakdown.sh r04_c4 3 --config c4 --eager > $O/r04v_profile_c4.log 2>&1; tail -2 $O/r04v_profile_c4.log
timeout -k 10 300 bash tools/kernel_breakdown.sh r04_c5 3 --config c5 > $O/r04v_profile_c5.log 2>&1; tail -2 $O/r04v_profile_c5.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04v_gen -- python3 tools/bench_generate.py --batch-beams > $O/r04v_gen.log 2>&1
cp $(find $O/r04v_gen -name '*_kernel_stats.csv' | head -1) $O/r04_generate_c3_kernel_stats.csv; rm -rf $O/r04v_gen
for R in 4 16; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04v_dec$R -- python3 tools/decode_step_probe.py --rows $R --steps 400 > $O/r04v_dec$R.log 2>&1
cp $(find $O/r04v_dec$R -name '*_kernel_stats.csv' | head -1) $O/r04_decode_step_rows${R}_kernel_stats.csv; rm -rf $O/r04v_dec$R
done
ls $O | grep "r04_" | head -40
