#!/usr/bin/env python3
"""Reduce rocprofv3 CSV output (kernel stats + PMC passes) to a small text/JSON summary.
usage: summarize_prof.py <prof_dir> <out_prefix>"""
import csv, glob, json, os, sys
from collections import defaultdict

prof, outp = sys.argv[1], sys.argv[2]
OURS = ("uhdr",)
summary = {}

def short(name):
    n = name.split("(")[0]
    return n[:110]

# kernel stats
rows = []
for f in glob.glob(os.path.join(prof, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append(r)
rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
# the product's kernels first; everything else in the process is torch generating the synthetic frames and the ceilings
# (libultrahdr_dev_amd/synth.py, scripts/mem_ceiling.py), outside the timed region: one aggregate line
ours_rows = [r for r in rows if any(o in r["Name"] for o in OURS)]
other_rows = [r for r in rows if not any(o in r["Name"] for o in OURS)]
summary["other_kernels_outside_the_timed_region"] = {"kernels": len(other_rows), "calls": sum(int(r["Calls"]) for r in other_rows),
                                                     "total_ms": round(sum(float(r["TotalDurationNs"]) for r in other_rows) / 1e6, 3)}
rows = ours_rows
summary["kernel_stats_top"] = [
    {"name": short(r["Name"]), "calls": int(r["Calls"]), "total_ns": int(float(r["TotalDurationNs"])),
     "avg_ns": float(r["AverageNs"]), "pct": float(r["Percentage"]), "min_ns": int(float(r["MinNs"])), "max_ns": int(float(r["MaxNs"]))}
    for r in rows[:12]]

# per-dispatch trace of our kernels: VGPR/SGPR/LDS + duration distribution
tr = defaultdict(list)
starts = defaultdict(list)
meta = {}
for f in glob.glob(os.path.join(prof, "stats", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if not any(o in n for o in OURS):
            continue
        tr[short(n)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        starts[short(n)].append(int(r["Start_Timestamp"]))
        meta[short(n)] = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                                                 "Workgroup_Size_X", "Grid_Size_X", "Grid_Size_Y")}
summary["our_kernels_trace"] = {n: {"dispatches": len(v), "avg_ns": sum(v) / len(v), "min_ns": min(v), "max_ns": max(v), **meta[n]}
                                for n, v in tr.items()}
# The timing pass runs `bench.py --steps 20 --warmup 5` with its 0.5 s clock ramp: the average over ALL dispatches includes the
# launches of the ramp, made while the card is still raising its clocks.  What bench.py times is the last 20 dispatches.
TIMED = 20
summary["timed_steps"] = {}
for n, v in tr.items():
    if len(v) > 2 * TIMED:
        last = [d for _, d in sorted(zip(starts[n], v))][-TIMED:]
        summary["timed_steps"][n] = {"dispatches": TIMED, "avg_ns": sum(last) / TIMED, "min_ns": min(last), "max_ns": max(last)}

# PMC passes: average per dispatch per kernel
for tag in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_clk"):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(prof, tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if not any(o in n for o in OURS):
                continue
            acc[short(n)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    summary[tag] = {n: {c: {"avg_per_dispatch": sum(v) / len(v), "dispatches": len(v)} for c, v in d.items()} for n, d in acc.items()}

# HBM traffic per launch of the two hot kernels.  FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies
# 128-B read requests at 64 B, so reads are doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.
traffic = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --steps 4 --warmup 1`; "
                     "bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 per launch of 64 4K frames"}
for key, pat in (("generate", "k_generate"), ("apply", "k_apply_s4")):
    f = [v["FETCH_SIZE"]["avg_per_dispatch"] for n, v in summary.get("pmc_fetch", {}).items() if pat in n and "FETCH_SIZE" in v]
    w = [v["WRITE_SIZE"]["avg_per_dispatch"] for n, v in summary.get("pmc_write", {}).items() if pat in n and "WRITE_SIZE" in v]
    if f and w:
        traffic[key] = int(2 * f[0] * 1024 + w[0] * 1024)
        traffic[key + "_read_bytes"] = int(2 * f[0] * 1024)
        traffic[key + "_write_bytes"] = int(w[0] * 1024)
import hashlib
_h = hashlib.sha256()
for _f in ("uhdr_kernels.hip", "uhdr_kernels.h", "uhdr_device_math.h"):
    _h.update(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "libultrahdr_dev_amd", "csrc", _f), "rb").read())
traffic["sources_sha16"] = _h.hexdigest()[:16]   # bench.py reports these bytes only while the kernels are the ones profiled
summary["traffic"] = traffic
# the JSON line the traced process itself printed: HIP events around the same launches the trace has timed (tracing serialises the
# kernels -- each dispatch begins when the one before has retired --, so a plain run of the same command can be faster than both:
# profiles/README.md, "the tracer and the boxes")
try:
    for line in open(os.path.join(prof, "stats.log")):
        if line.startswith("{"):
            d = json.loads(line)
            summary["traced_process_bench_line"] = {"value": d["value"], "ms_per_step": d["ms_per_step"],
                                                    "generate_avg_launch_ms": d["kernels"]["generate"]["avg_launch_ms"],
                                                    "apply_avg_launch_ms": d["kernels"]["apply"]["avg_launch_ms"]}
except Exception:
    pass
json.dump(traffic, open(outp + "_traffic.json", "w"), indent=1)
json.dump(summary, open(outp + ".json", "w"), indent=1)
with open(outp + ".txt", "w") as o:
    o.write("== rocprofv3 --kernel-trace --stats: the product's kernels (everything else in the process -- torch kernels that synthesise the frames, outside the timed region: %s) ==\n" % json.dumps(summary["other_kernels_outside_the_timed_region"]))
    for k in summary["kernel_stats_top"]:
        o.write("%-112s calls=%-6d avg=%10.1f us  total=%10.3f ms  %5.1f%%\n" % (k["name"], k["calls"], k["avg_ns"] / 1e3, k["total_ns"] / 1e6, k["pct"]))
    o.write("\n== the last %d dispatches of each hot kernel = the K timed steps of `bench.py --steps 20 --warmup 5` (the averages above include the ~450 launches of the 0.5 s clock ramp, made while the clocks are still rising) ==\n" % TIMED)
    for n, v in summary["timed_steps"].items():
        o.write("%-112s avg=%10.1f us  min=%8.1f  max=%8.1f\n" % (n, v["avg_ns"] / 1e3, v["min_ns"] / 1e3, v["max_ns"] / 1e3))
    if "traced_process_bench_line" in summary:
        o.write("\n== the bench line of this traced process (HIP events in the same run) ==\n%s\n" % json.dumps(summary["traced_process_bench_line"]))
    o.write("\n== our kernels (per dispatch) ==\n")
    for n, v in summary["our_kernels_trace"].items():
        o.write("%s\n   %s\n" % (n, json.dumps(v)))
    for tag in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_clk"):
        o.write("\n== %s (avg per dispatch) ==\n" % tag)
        for n, d in summary[tag].items():
            o.write("%s\n" % n)
            for c, v in d.items():
                o.write("   %-24s %18.1f  (n=%d)\n" % (c, v["avg_per_dispatch"], v["dispatches"]))
    o.write("\n== HBM traffic per launch ==\n%s\n" % json.dumps(traffic, indent=1))
print(open(outp + ".txt").read())
