#!/usr/bin/env python3
"""The bench step (64 x 4K: generate, then apply) over ROTATING resident batches (step k works on batch k mod R, so no step finds
what the step before it left in the 256 MB Infinity Cache), issued

  * as the bench issues it: one 64-frame generate launch, one 64-frame apply launch, one stream;
  * in chunks of C frames -- generate(chunk), apply(chunk) -- on one stream;
  * in chunks alternating over S streams, so that chunk c's apply runs beside chunk c+1's generate and each chunk's ramps and
    k_generate_resolve latency hide behind the other stream's kernels.

The point of the chunks: apply re-reads the 12.4 MB of 8-bit YUV per frame that generate read a moment
before (15 % of the step's bytes); with C x 12.4 MB under the cache's size and generate loading those planes with plain loads
(library variant GT of scripts/ab/policy_variants.py) the second read may be served on-die.

    UHDR_HIP_LIB=scripts/ab/libvar_GT.so python scripts/time_step_pipeline.py [R] -> one line per (chunk, streams): ms per step

Output: ms per 64-frame step, median of `REPS` repeats of `STEPS` steps each, for every configuration, interleaved.
"""
import ctypes as C, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from libultrahdr_dev_amd import api
torch.cuda.set_device(0)
lib = api.init(0)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 3
STEPS = int(os.environ.get("STEPS", "60"))
REPS = int(os.environ.get("REPS", "3"))
CONFIGS = [tuple(int(x) for x in c.split("x")) for c in os.environ.get("CONFIGS", "64x1,32x2,16x1,16x2,8x1,8x2,8x3,4x2,4x4").split(",")]
batches = [bench.Batch(lib, 64, 0, seed_offset=4096 * r) for r in range(R)]
fmt = api.OUTPUT_HDR_HLG
streams = [torch.cuda.Stream() for _ in range(4)]
hs = [C.c_void_p(s.cuda_stream) for s in streams]
md = api.Metadata()


def step(b, k, ns, base):
    """one pass over batch b in chunks of k frames, chunk c on stream (base + c) % ns"""
    c = base
    for lo in range(0, 64, k):
        s = hs[c % ns]
        rc = lib.uhdr_hip_generate_gainmap_batch(k, b._slice(b.yi, lo), b._slice(b.pi, lo), api.TF_HLG, C.byref(md), b._slice(b.mi, lo), 0,
                                                 C.c_void_p(b.minmax.data_ptr() + 8 * lo), s)
        assert rc == 0
        rc = lib.uhdr_hip_apply_gainmap_batch(k, b._slice(b.yi, lo), b._slice(b.mi, lo), C.byref(md), fmt, api.FLT_MAX, b._slice(b.oi, lo), api.APPLY_FAST, s)
        assert rc == 0
        c += 1
    return c


def run(k, ns, steps):
    base = 0
    for i in range(6):
        base = step(batches[i % R], k, ns, base)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        base = step(batches[i % R], k, ns, base)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for _ in range(300):   # clocks
    step(batches[0], 64, 1, 0)
torch.cuda.synchronize()
res = {c: [] for c in CONFIGS}
for rep in range(REPS):
    for (k, ns) in CONFIGS:
        res[(k, ns)].append(run(k, ns, STEPS))
print("lib", os.environ.get("UHDR_HIP_LIB", "shipped"), "rotating over", R, "batches")
for (k, ns) in CONFIGS:
    v = res[(k, ns)]
    ms = statistics.median(v)
    print("chunk %2d x %d stream(s): %.4f ms per step (%s)  %.0f MPix/s" % (k, ns, ms, " ".join("%.4f" % x for x in v), 64 * 3840 * 2160 / ms / 1e3))
