#!/usr/bin/env python3
"""the bench step (64 x 4K: generate, then apply) issued in chunks -- generate(k frames), apply(the same k), next k ... -- so that the
frames apply reads again (the 8-bit YUV, 12.4 MB each) may still be in the 256 MB Infinity Cache: ms per 64-frame step by chunk size.
Round 3, one box: 64 -> 0.999-1.011 ms, 32 -> 1.052, 16 -> 1.239, 8 -> 1.384, 4 -> 2.224: nothing comes back from the cache that would pay for the
shorter launches (each chunk has its own k_generate_resolve latency and its own ramps)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from libultrahdr_dev_amd import api
torch.cuda.set_device(0)
lib = api.init(0)
b = bench.Batch(lib, 64, 0)
fmt = api.OUTPUT_HDR_HLG
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
md = api.Metadata()


def step(k):
    for lo in range(0, 64, k):
        rc = lib.uhdr_hip_generate_gainmap_batch(k, b._slice(b.yi, lo), b._slice(b.pi, lo), api.TF_HLG, C.byref(md), b._slice(b.mi, lo), 0,
                                                 C.c_void_p(b.minmax.data_ptr() + 8 * lo), s)
        assert rc == 0
        rc = lib.uhdr_hip_apply_gainmap_batch(k, b._slice(b.yi, lo), b._slice(b.mi, lo), C.byref(md), fmt, api.FLT_MAX, b._slice(b.oi, lo), api.APPLY_FAST, s)
        assert rc == 0


for _ in range(300):
    step(64)
torch.cuda.synchronize()
for rep in range(3):
    out = []
    for k in (64, 32, 16, 8, 4):
        for _ in range(5):
            step(k)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40):
            step(k)
        torch.cuda.synchronize()
        out.append("chunk %2d: %.4f ms" % (k, (time.perf_counter() - t0) / 40 * 1e3))
    print(" | ".join(out))
