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
summary["kernel_stats_top"] = [
    {"name": short(r["Name"]), "calls": int(r["Calls"]), "total_ns": int(float(r["TotalDurationNs"])),
     "avg_ns": float(r["AverageNs"]), "pct": float(r["Percentage"]), "min_ns": int(float(r["MinNs"])), "max_ns": int(float(r["MaxNs"]))}
    for r in rows[:12]]

# per-dispatch trace of our kernels: VGPR/SGPR/LDS + duration distribution
tr = defaultdict(list)
meta = {}
for f in glob.glob(os.path.join(prof, "stats", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if not any(o in n for o in OURS):
            continue
        tr[short(n)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        meta[short(n)] = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                                                 "Workgroup_Size_X", "Grid_Size_X", "Grid_Size_Y")}
summary["our_kernels_trace"] = {n: {"dispatches": len(v), "avg_ns": sum(v) / len(v), "min_ns": min(v), "max_ns": max(v), **meta[n]}
                                for n, v in tr.items()}

# PMC passes: average per dispatch per kernel
for tag in ("pmc_fetch", "pmc_write", "pmc_sq"):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(prof, tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if not any(o in n for o in OURS):
                continue
            acc[short(n)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    summary[tag] = {n: {c: {"avg_per_dispatch": sum(v) / len(v), "dispatches": len(v)} for c, v in d.items()} for n, d in acc.items()}

json.dump(summary, open(outp + ".json", "w"), indent=1)
with open(outp + ".txt", "w") as o:
    o.write("== rocprofv3 --kernel-trace --stats: top kernels ==\n")
    for k in summary["kernel_stats_top"]:
        o.write("%-112s calls=%-6d avg=%10.1f us  total=%10.3f ms  %5.1f%%\n" % (k["name"], k["calls"], k["avg_ns"] / 1e3, k["total_ns"] / 1e6, k["pct"]))
    o.write("\n== our kernels (per dispatch) ==\n")
    for n, v in summary["our_kernels_trace"].items():
        o.write("%s\n   %s\n" % (n, json.dumps(v)))
    for tag in ("pmc_fetch", "pmc_write", "pmc_sq"):
        o.write("\n== %s (avg per dispatch) ==\n" % tag)
        for n, d in summary[tag].items():
            o.write("%s\n" % n)
            for c, v in d.items():
                o.write("   %-24s %18.1f  (n=%d)\n" % (c, v["avg_per_dispatch"], v["dispatches"]))
print(open(outp + ".txt").read())
