#!/bin/bash
# per-kernel average durations of a short bench run: scripts/kernel_times.sh [bench args]  -> stdout
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/ktimes_$$
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 6 --warmup 2 --ramp-ms 0 --no-cpu-baseline --no-other-configs "$@" > $OUT/log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "uhdr" in r["Name"] or float(r["Percentage"]) > 2:
            print("%-100s calls=%-5s avg=%9.1f us  %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
rm -rf $OUT
