#!/bin/bash
# kernel timeline (start, gap in front, duration) of a slice of a script's kernel trace: TRACE_SCRIPT, TRACE_FROM / TRACE_TO = substrings of the
# kernel names that open / close the slice, TRACE_SKIP = how many openers to skip
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/trace_calls
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/a -- python3 $R/scripts/${TRACE_SCRIPT:-time_jpegr.py} > $OUT/a.log 2>&1
python3 - $OUT $R/gpurun_out/trace_calls.txt <<'PY'
import csv, glob, os, sys
f = glob.glob(sys.argv[1] + "/a/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
opener = os.environ.get("TRACE_FROM", "k_generate")
starts = [i for i, r in enumerate(rows) if opener in r["Kernel_Name"]]
k = int(os.environ.get("TRACE_SKIP", "8"))
a, b = starts[k], starts[k + 1]
out = open(sys.argv[2], "w")
t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.write("%8.1f us  +%6.1f gap  %6.1f us  %s\n" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:80]))
    prev_end = e
out.write("next opener %.1f us after the last kernel\n" % ((int(rows[b]["Start_Timestamp"]) - prev_end) / 1e3))
PY
cat $R/gpurun_out/trace_calls.txt
