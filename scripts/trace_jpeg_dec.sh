#!/bin/bash
# timeline of one decode: start offset, duration and the gap in front of every kernel of the last quality-95 decode of scripts/time_jpeg_dec.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/trace_jd
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/a -- python3 $R/scripts/${TRACE_SCRIPT:-time_jpeg_dec.py} > $OUT/a.log 2>&1
python3 - $OUT $R/gpurun_out/trace_jpeg_dec.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/a/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_jd_prepare_multi" in r["Kernel_Name"]]
# decodes 4..13 are the 10 timed q95 decodes; take the 8th
import os
k = int(os.environ.get("TRACE_DECODE", "10"))
a, b = starts[k], starts[k + 1]
out = open(sys.argv[2], "w")
t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.write("%8.1f us  +%6.1f gap  %6.1f us  %s\n" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:70]))
    prev_end = e
out.write("total span %.1f us; next decode starts %.1f us after this one's last kernel\n" % ((prev_end - t0) / 1e3, (int(rows[b]["Start_Timestamp"]) - prev_end) / 1e3))
PY
cat $R/gpurun_out/trace_jpeg_dec.txt
