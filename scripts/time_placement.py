#!/usr/bin/env python3
"""Does WHERE a resident batch lies in device memory matter?  One process, eight batches allocated one after another (all kept, so
each lies elsewhere), the step timed on each in turn, three rounds, windows of 40 steps without events; per batch: the arenas' base
addresses and the step's time.  Differences between batches that repeat round after round are placement; differences between
rounds are the card."""
import ctypes as C, os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from libultrahdr_dev_amd import api
torch.cuda.set_device(0)
lib = api.init(0)
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 8
batches = [bench.Batch(lib, 64, 0, seed_offset=65536 * r) for r in range(NB)]
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
fmt = api.OUTPUT_HDR_HLG


def window(b, n=40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        b.generate(s); b.apply(s, fmt)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def kernel_ms(b, which, n=40):
    f = (lambda: b.generate(s)) if which == "g" else (lambda: b.apply(s, fmt))
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for _ in range(20): window(batches[0])
res = [[] for _ in batches]
ga = [[[], []] for _ in batches]
for rnd in range(3):
    for i, b in enumerate(batches):
        res[i].append(window(b))
    for i, b in enumerate(batches):
        ga[i][0].append(kernel_ms(b, "g")); ga[i][1].append(kernel_ms(b, "a"))
for i, b in enumerate(batches):
    print("batch %d  p010 %#x yuv %#x map %#x out %#x   step %s ms   generate alone %s   apply alone %s" % (
        i, b.p010[0].data_ptr(), b.yuv[0].data_ptr(), b.maps[0].data_ptr(), b.outs[0].data_ptr(),
        " ".join("%.4f" % v for v in res[i]), " ".join("%.4f" % v for v in ga[i][0]), " ".join("%.4f" % v for v in ga[i][1])))
