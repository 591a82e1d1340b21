#!/bin/bash
# effective shader clock during the decoder's kernels: GRBM_GUI_ACTIVE (cycles the GPU was busy, per XCD) / kernel duration
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/prof_jclk
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/a -- python3 $R/scripts/time_jpeg_dec.py > $OUT/a.log 2>&1
python3 - $OUT $R/gpurun_out/prof_jpeg_clock.txt <<'PY'
import csv, glob, sys, collections
cnt = glob.glob(sys.argv[1] + "/a/**/*counter_collection.csv", recursive=True)
out = open(sys.argv[2], "w")
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for f in cnt:
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != "GRBM_GUI_ACTIVE":
            continue
        name = r["Kernel_Name"][:60]
        dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) if "End_Timestamp" in r else 0.0
        a = agg[name]
        a[0] += float(r["Counter_Value"]); a[1] += dur; a[2] += 1
for k, (c, d, n) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if d > 0:
        out.write("%-62s n=%-5d avg %.1f us  GRBM_GUI_ACTIVE/8 per ns = %.2f GHz\n" % (k, n, d / n / 1e3, c / 8.0 / d))
PY
cat $R/gpurun_out/prof_jpeg_clock.txt | head -12
