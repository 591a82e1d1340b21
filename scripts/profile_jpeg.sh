#!/bin/bash
# rocprofv3 --kernel-trace --stats over the device JPEG encoder / decoder timing scripts -> gpurun_out/prof_jpeg_<tag>.txt
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-run}
OUT=/tmp/prof_jpeg_$TAG
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/enc -- python3 $R/scripts/time_jpeg.py > $OUT/enc.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dec -- python3 $R/scripts/time_jpeg_dec.py > $OUT/dec.log 2>&1
python3 - $OUT $R/gpurun_out/prof_jpeg_$TAG.txt <<'PY'
import csv, glob, sys
out = open(sys.argv[2], "w")
for name, title in (("enc", "uhdr_hip_jpeg_encode, one smooth 4K YUV420 frame, q95 and q85 (scripts/time_jpeg.py)"),
                    ("dec", "uhdr_hip_jpeg_decode of those files (scripts/time_jpeg_dec.py)")):
    out.write("== %s ==\n" % title)
    for l in open("%s/%s.log" % (sys.argv[1], name)):
        if "us per" in l:
            out.write("   (under the profiler) " + l)
    f = glob.glob("%s/%s/**/*kernel_stats.csv" % (sys.argv[1], name), recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if "uhdr::" in r["Name"] or "rocprim" in r["Name"] or "rocclr" in r["Name"]:
            out.write("%-100s calls=%-5s avg=%8.1f us  total=%8.2f ms\n" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
    out.write("\n")
PY
rm -rf $OUT
cat $R/gpurun_out/prof_jpeg_$TAG.txt
