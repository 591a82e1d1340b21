#!/bin/bash
# SQ / LDS counters of the apply kernel over a short bench run -> gpurun_out/pmc_apply_<tag>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-run}
OUT=/tmp/pmc_apply_$TAG
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 4 --warmup 1 --ramp-ms 0 --no-fixed-batch --no-placement-ab --no-cpu-baseline --no-other-configs"
timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 $ARGS > $OUT/a.log 2>&1 || echo "pass a: timed out or failed" >> $R/gpurun_out/pmc_apply_progress_$TAG.log
echo "pass a done $(date +%T)" >> $R/gpurun_out/pmc_apply_progress_$TAG.log
timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/b -- python3 $ARGS > $OUT/b.log 2>&1 || echo "pass b: timed out or failed" >> $R/gpurun_out/pmc_apply_progress_$TAG.log
echo "pass b done $(date +%T)" >> $R/gpurun_out/pmc_apply_progress_$TAG.log
timeout -k 10 150 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $OUT/c -- python3 $ARGS > $OUT/c.log 2>&1 || echo "pass c: timed out or failed" >> $R/gpurun_out/pmc_apply_progress_$TAG.log
echo "pass c done $(date +%T)" >> $R/gpurun_out/pmc_apply_progress_$TAG.log
# instruction classes that would show scratch traffic (scratch_* count as FLAT: FLAT must equal VMEM_RD + VMEM_WR).  The L2 counters
# (TCC_HIT_sum / TCC_MISS_sum) come from scripts/profile_bench.sh; a pass with TCC_REQ / TCC_EA0_* crashed rocprofv3 on this image.
timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d $OUT/e -- python3 $ARGS > $OUT/e.log 2>&1 || echo "pass e: timed out or failed" >> $R/gpurun_out/pmc_apply_progress_$TAG.log
echo "pass e done $(date +%T)" >> $R/gpurun_out/pmc_apply_progress_$TAG.log
python3 - $OUT $R/gpurun_out/pmc_apply_$TAG.txt <<'PY'
import csv, glob, sys, collections
out = open(sys.argv[2], "w")
for d in "abce":
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (sys.argv[1], d), recursive=True):
        for r in csv.DictReader(open(f)):
            if "uhdr::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in acc:
        out.write(k + "\n")
        for c, v in sorted(acc[k].items()):
            out.write("   %-28s %16.1f (n=%d)\n" % (c, sum(v) / len(v), len(v)))
    if not acc:
        out.write("pass %s: no data\n%s\n" % (d, open("%s/%s.log" % (sys.argv[1], d)).read()[-1500:]))
PY
rm -rf $OUT
cat $R/gpurun_out/pmc_apply_$TAG.txt
