#!/usr/bin/env python3
"""one process, no events: windows of 60 steps alternating between rotation over 3 batches and one fixed batch -- what rotation
costs on this box, free of any order-in-process effect"""
import ctypes as C, os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from libultrahdr_dev_amd import api
torch.cuda.set_device(0)
lib = api.init(0)
batches = [bench.Batch(lib, 64, 0, seed_offset=65536 * r) for r in range(3)]
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
fmt = api.OUTPUT_HDR_HLG
k = 0
def window(rot, n=60):
    global k
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        b = batches[k % 3] if rot else batches[0]; k += 1
        b.generate(s); b.apply(s, fmt)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for _ in range(10): window(True)
res = {True: [], False: []}
for i in range(16):
    for rot in (True, False):
        res[rot].append(window(rot))
for rot in (True, False):
    v = res[rot]
    print("%-28s median %.4f ms (min %.4f max %.4f) = %.0f MPix/s" % ("rotating over 3 batches" if rot else "one fixed batch", statistics.median(v), min(v), max(v), 64 * 3840 * 2160 / statistics.median(v) / 1e3))
