#!/usr/bin/env python3
"""How the step time develops under sustained load: the 64 x 4K step (rotating over R batches) issued continuously for SECONDS,
ms per step over consecutive windows of 40 steps.  Answers what `bench.py --ramp-ms` should be: how long after an idle card starts
working its rate is the rate it then keeps.    python scripts/time_step_course.py [R] [seconds]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from libultrahdr_dev_amd import api
torch.cuda.set_device(0)
lib = api.init(0)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 3
SECONDS = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
batches = [bench.Batch(lib, 64, 0, seed_offset=65536 * r) for r in range(R)]
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
fmt = api.OUTPUT_HDR_HLG
torch.cuda.synchronize()
time.sleep(1.0)   # an idle card
t_start = time.perf_counter()
k, rows = 0, []
while time.perf_counter() - t_start < SECONDS:
    t0 = time.perf_counter()
    for _ in range(40):
        b = batches[k % R]; k += 1
        b.generate(s); b.apply(s, fmt)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    rows.append((t0 - t_start, (t1 - t0) / 40 * 1e3))
print("R = %d; seconds since the first launch : ms per step (40-step windows)" % R)
for i, (t, ms) in enumerate(rows):
    if i < 12 or i % 10 == 0:
        print("%7.3f s  %.4f ms  %.0f MPix/s" % (t, ms, 64 * 3840 * 2160 / ms / 1e3))
import statistics
for lo, hi in ((0.0, 0.5), (0.5, 1.0), (1.0, 2.0), (2.0, 4.0), (4.0, 8.0), (8.0, 1e9)):
    v = [ms for t, ms in rows if lo <= t < hi]
    if v:
        print("window %4.1f-%4.1f s: median %.4f ms = %.0f MPix/s" % (lo, min(hi, SECONDS), statistics.median(v), 64 * 3840 * 2160 / statistics.median(v) / 1e3))
