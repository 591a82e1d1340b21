#!/bin/bash
# usage: profile_pmc.sh <tag> ; PMC passes (SQ mix + memory) over bench, summary -> gpurun_out/pmc_<tag>.txt
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-x}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1 || true
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1 || true
python3 $R/scripts/summarize_prof.py $OUT $R/gpurun_out/pmc_$TAG > /dev/null
rm -rf $OUT
cat $R/gpurun_out/pmc_$TAG.txt
