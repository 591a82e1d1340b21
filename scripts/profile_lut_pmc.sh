#!/bin/bash
# SQ / LDS counters of the LUT-mode apply kernel (scripts/time_lut.py) -> gpurun_out/pmc_lut_<tag>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-run}
OUT=/tmp/pmc_lut_$TAG
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="$R/scripts/time_lut.py"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 $ARGS > $OUT/a.log 2>&1 || echo "pass a: timed out or failed" >> $R/gpurun_out/pmc_lut_progress_$TAG.log
echo "pass a done $(date +%T)" >> $R/gpurun_out/pmc_lut_progress_$TAG.log
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/b -- python3 $ARGS > $OUT/b.log 2>&1 || echo "pass b: timed out or failed" >> $R/gpurun_out/pmc_lut_progress_$TAG.log
echo "pass b done $(date +%T)" >> $R/gpurun_out/pmc_lut_progress_$TAG.log
python3 - $OUT $R/gpurun_out/pmc_lut_$TAG.txt <<'PY'
import csv, glob, sys, collections
out = open(sys.argv[2], "w")
for d in "ab":
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (sys.argv[1], d), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_apply_lut" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:60] + " grid " + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in acc:
        out.write(k + "\n")
        for c, v in sorted(acc[k].items()):
            out.write("   %-28s %16.1f (n=%d)\n" % (c, sum(v) / len(v), len(v)))
    if not acc:
        out.write("pass %s: no data\n%s\n" % (d, open("%s/%s.log" % (sys.argv[1], d)).read()[-1500:]))
PY
rm -rf $OUT
cat $R/gpurun_out/pmc_lut_$TAG.txt
