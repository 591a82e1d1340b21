#!/bin/bash
# rocprofv3 passes over the bench command (run on the GPU box from the repo root):
#   1) --kernel-trace --stats   2) --pmc FETCH_SIZE   3) --pmc WRITE_SIZE   4) SQ instruction mix   5) clocks / L2
# Counters are collected in their own runs (never combined with trace domains other than kernel-trace).
# usage: profile_bench.sh <tag>   -> gpurun_out/prof_<tag>.{txt,json} (+ traffic json)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-run}
OUT=$R/gpurun_out/prof_raw_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the timing pass runs the bench as the driver does (clock ramp included: its kernel averages are the ones to agree with the bench
# line); counters do not depend on clocks, so their passes skip the ramp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-fixed-batch --no-placement-ab --no-cpu-baseline --no-other-configs > $OUT/stats.log 2>&1
ARGS="$R/bench.py --steps 4 --warmup 1 --ramp-ms 0 --no-fixed-batch --no-placement-ab --no-cpu-baseline --no-other-configs"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_clk -- python3 $ARGS > $OUT/pmc_clk.log 2>&1 || true
python3 $R/scripts/summarize_prof.py $OUT $R/gpurun_out/prof_$TAG
rm -rf $OUT
