#!/bin/bash
# per-kernel average durations of any python script: scripts/kernel_times_of.sh scripts/time_exact.py [args]  -> stdout
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/ktimes_$$
rm -rf $OUT; mkdir -p $OUT
S=$R/$1; shift
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $S "$@" > $OUT/log 2>&1
tail -n 20 $OUT/log
python3 - $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "uhdr" in r["Name"] or float(r["Percentage"]) > 2:
            print("%-110s calls=%-5s avg=%9.1f us  %5.1f%%" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
rm -rf $OUT
