#!/usr/bin/env python3
"""latency of the single-image FAST apply calls of BASELINE configs[4] and configs[1]: 8K -> PQ RGBA1010102 / RGBA F16, 4K -> HLG; one line
(microseconds per call, back-to-back calls on one stream)"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from libultrahdr_dev_amd import api, synth
lib = api.init(0)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn, iters=60):
    for _ in range(8):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


out = {}
for (W, H) in ((7680, 4320), (3840, 2160)):
    _, y = synth.lcg_frame(W, H, 1234)
    m = torch.randint(0, 256, ((W // 4) * (H // 4),), dtype=torch.uint8, device="cuda")
    o = torch.zeros(W * H * 8, dtype=torch.uint8, device="cuda")
    yi, mi, oi = api.yuv420_image(y.data_ptr(), W, H, api.CG_BT709), api.mono_image(m.data_ptr(), W // 4, H // 4), api.out_image(o.data_ptr())
    md = api.metadata(float(np.float32(10000.0) / np.float32(203.0)))
    for name, fmt in (("pq", api.OUTPUT_HDR_PQ), ("hlg", api.OUTPUT_HDR_HLG), ("f16", api.OUTPUT_HDR_LINEAR)):
        out["%dx%d_%s_us" % (W, H, name)] = round(timed(lambda: lib.uhdr_hip_apply_gainmap(C.byref(yi), C.byref(mi), C.byref(md), fmt, api.FLT_MAX, C.byref(oi),
                                                                                         api.APPLY_FAST, api.MEM_DEVICE, stream)), 2)
print(json.dumps(out))
