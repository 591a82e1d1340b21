#!/usr/bin/env python3
"""EXACT apply (the reference's bytes), 32 x 4K per call -> HLG: ms per call (run under scripts/kernel_times_of.sh for the kernels)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from libultrahdr_dev_amd import api
torch.cuda.set_device(0)
lib = api.init(0)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
b = bench.Batch(lib, n, 0)
b.generate(stream)
for fmt, name in ((api.OUTPUT_HDR_HLG, "HLG"), (api.OUTPUT_HDR_PQ, "PQ")):
    def f():
        rc = lib.uhdr_hip_apply_gainmap_batch(n, b.yi, b.mi, C.byref(b.md), fmt, api.FLT_MAX, b.oi, api.APPLY_EXACT, stream)
        assert rc == 0, rc
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("EXACT %d x 4K -> %s: %.4f ms per call, %.1f us per frame" % (n, name, ms, ms * 1e3 / n))
