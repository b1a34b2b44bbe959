#!/bin/bash
# rocprofv3 evidence for one build, run ON THE GPU BOX (gpurun):  tools/profile_round.sh r02a [bench args]
# Separate passes (the guide's HBM/rocprofv3 section: PMC counters in their own runs, with
# --kernel-trace only): kernel stats, FETCH_SIZE, WRITE_SIZE, two SQ/GRBM sets.
# The program itself follows `--` (python3 bench.py ...), never a shell wrapper.
set -e -o pipefail
TAG=${1:-r02}; shift || true
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-side-configs $*"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 bench.py $ARGS > $OUT/${TAG}_stats.log 2>&1
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 bench.py $ARGS > $OUT/${TAG}_pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- python3 bench.py $ARGS > $OUT/${TAG}_pmc_write.log 2>&1
echo "write done"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_sq1 -- python3 bench.py $ARGS > $OUT/${TAG}_pmc_sq1.log 2>&1
echo "sq1 done"
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES --output-format csv -d $OUT/${TAG}_pmc_sq2 -- python3 bench.py $ARGS > $OUT/${TAG}_pmc_sq2.log 2>&1
echo "sq2 done"
# the raw per-dispatch CSVs are large; keep only what profiles/summarize.py reads
find $OUT/${TAG}_* -name '*_agent_info.csv' -delete 2>/dev/null || true
python3 profiles/summarize.py $TAG --on-box $*
ls -la $OUT/${TAG}_summary
