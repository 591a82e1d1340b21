#!/bin/bash
# one box: the bench plain, then under rocprofv3 --kernel-trace (its own JSON line = events in the traced process, and the trace's
# per-kernel averages over the last 20 dispatches), then plain again -> stdout
R=${GRAFT_REPO_ROOT:-$(pwd)}
P='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], "value", d["value"], "ms_per_step", d["ms_per_step"], "generate", d["kernels"]["generate"]["avg_launch_ms"], "apply", d["kernels"]["apply"]["avg_launch_ms"])'
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-other-configs"
python3 $R/bench.py $ARGS 2>/dev/null | python3 -c "$P" plain
OUT=/tmp/tve_$$; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py $ARGS > $OUT/log 2>&1 )
python3 -c "$P" traced_process_events < $OUT/log
python3 - $OUT <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "uhdr::k_generate" in r["Kernel_Name"] or "uhdr::k_apply" in r["Kernel_Name"]:
            rows[r["Kernel_Name"][:40]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
allk = sorted((s, e, k) for k, v in rows.items() for s, e in v)
for k, v in rows.items():
    v.sort(); last = v[-20:]
    print("trace", k, "avg of last 20: %.1f us" % (sum(e - s for s, e in last) / len(last) / 1e3))
last60 = allk[-60:]
gaps = [last60[i + 1][0] - last60[i][1] for i in range(len(last60) - 1)]
print("trace: span of the last 20 steps %.1f us per step; gaps between consecutive kernels avg %.2f us (min %.2f, max %.2f)" % (
    (last60[-1][1] - last60[0][0]) / 20e3, sum(gaps) / len(gaps) / 1e3, min(gaps) / 1e3, max(gaps) / 1e3))
PY
rm -rf $OUT
python3 $R/bench.py $ARGS 2>/dev/null | python3 -c "$P" plain
