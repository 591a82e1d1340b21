#!/usr/bin/env python3
"""Every fraction of the bench line, recomputed from files of ONE box:
    python scripts/recompute_fractions.py gpurun_out/bench_TAG.json gpurun_out/prof_TAG.json
bench_TAG.json: the plain `python bench.py` line; prof_TAG.json: scripts/profile_bench.sh's summary of the same box (kernel trace of
`bench.py --steps 20 --warmup 5`, the JSON line that traced process printed, the PMC passes)."""
import json
import sys

W, H, N = 3840, 2160, 64
GEN = W * H * 3 + W * H * 3 // 2 + (W // 4) * (H // 4)
APP = W * H * 3 // 2 + (W // 4) * (H // 4) + W * H * 4
PEAK = 8.0e12
b = json.load(open(sys.argv[1]))
p = json.load(open(sys.argv[2]))
out = []


def line(what, num_bytes, seconds, src):
    out.append("%-78s %14d B / %9.4f ms = %6.3f TB/s = %.4f of 8 TB/s   [%s]" % (what, num_bytes, seconds * 1e3, num_bytes / seconds / 1e12, num_bytes / seconds / PEAK, src))


ms = b["ms_per_step"] * 1e-3
line("whole step, plain run (the number to quote: roofline.frac_whole_step)", (GEN + APP) * N, ms, "bench line: ms_per_step")
assert abs((GEN + APP) * N / ms / PEAK - b["roofline"]["frac_whole_step"]) < 2e-4
if b.get("fixed_batch"):
    line("whole step, plain run, FIXED batch (rounds 1-3's protocol)", (GEN + APP) * N, b["fixed_batch"]["ms_per_step"] * 1e-3, "bench line: fixed_batch.ms_per_step")
if b.get("cold_start"):
    line("whole step, plain run, card as the setup leaves it", (GEN + APP) * N, b["cold_start"]["ms_per_step"] * 1e-3, "bench line: cold_start.ms_per_step")
line("apply, HIP events around each launch, plain run (roofline.frac)", APP * N, b["kernels"]["apply"]["avg_launch_ms"] * 1e-3, "bench line: kernels.apply.avg_launch_ms")
line("generate (+ resolve), HIP events, plain run", GEN * N, b["kernels"]["generate"]["avg_launch_ms"] * 1e-3, "bench line: kernels.generate.avg_launch_ms")
t = p.get("traced_process_bench_line")
if t:
    line("whole step, the TRACED process's own line", (GEN + APP) * N, t["ms_per_step"] * 1e-3, "summary: traced_process_bench_line.ms_per_step")
    line("apply, HIP events in the traced process", APP * N, t["apply_avg_launch_ms"] * 1e-3, "summary: traced_process_bench_line")
ksum = 0.0
for name, v in p.get("timed_steps", {}).items():
    if "k_apply_s4<" in name:
        line("apply, rocprofv3 kernel trace, the 20 timed dispatches: " + name.split("(")[0][-28:], APP * N, v["avg_ns"] * 1e-9, "summary: timed_steps")
    if "k_generate<" in name:
        line("generate kernel alone, kernel trace, the 20 timed dispatches", GEN * N, v["avg_ns"] * 1e-9, "summary: timed_steps")
    if any(k in name for k in ("k_apply_s4<", "k_generate<", "k_generate_resolve<")):
        ksum += v["avg_ns"] * 1e-9
if t and ksum:
    out.append("sum of the three kernels' traced averages %.4f ms  <=  the traced process's ms_per_step %.4f ms: %s" % (ksum * 1e3, t["ms_per_step"], ksum * 1e3 <= t["ms_per_step"] + 1e-3))
    out.append("plain ms_per_step %.4f ms against the traced kernels' sum %.4f ms: in a plain run consecutive kernels overlap at their boundaries by %.1f us per step"
               % (b["ms_per_step"], ksum * 1e3, (ksum * 1e3 - b["ms_per_step"]) * 1e3))
tr = p.get("traffic", {})
for k, alg in (("apply", APP * N), ("generate", GEN * N)):
    if k in tr:
        out.append("HBM traffic of %-8s per launch (PMC: 2 x FETCH_SIZE + WRITE_SIZE, KiB): %d B = %.4f x the algorithmic %d B" % (k, tr[k], tr[k] / alg, alg))
print("\n".join(out))
