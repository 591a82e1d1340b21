#!/bin/bash
# rocprofv3 --kernel-trace --stats over scripts/time_jpegr.py -> gpurun_out/prof_jpegr_<tag>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-run}
OUT=/tmp/prof_jpegr_$TAG
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/a -- python3 $R/scripts/time_jpegr.py > $OUT/a.log 2>&1
python3 - $OUT $R/gpurun_out/prof_jpegr_$TAG.txt <<'PY'
import csv, glob, sys
out = open(sys.argv[2], "w")
out.write("== uhdr_hip_jpegr_encode_api0 / api1 / uhdr_hip_jpegr_decode, 4K and 640x480 (scripts/time_jpegr.py: 23 calls of each per size) ==\n")
for l in open(sys.argv[1] + "/a.log"):
    if "encode API" in l:
        out.write("   (under the profiler) " + l)
f = glob.glob(sys.argv[1] + "/a/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "uhdr::" in r["Name"] or "rocprim" in r["Name"] or "rocclr" in r["Name"]:
        out.write("%-100s calls=%-5s avg=%8.1f us  total=%8.2f ms\n" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
rm -rf $OUT
cat $R/gpurun_out/prof_jpegr_$TAG.txt
