#!/bin/bash
# rocprofv3 passes over the bench command (run on the GPU box from the repo root).
# 1) --kernel-trace --stats  2) --pmc FETCH_SIZE  3) --pmc WRITE_SIZE  4) SQ instruction mix
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 4 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
find $OUT -name "*.csv" | head -50
python3 $R/scripts/summarize_prof.py $OUT $R/gpurun_out/prof_summary
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
