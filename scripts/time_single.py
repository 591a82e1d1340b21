#!/usr/bin/env python3
"""latency of single-image calls (BASELINE configs[1]): one 4K HLG generate, one 4K apply -> HLG, 8K apply -> PQ"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from libultrahdr_dev_amd import api, synth

lib = api.init(0)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


out = {}
for (W, H) in ((3840, 2160), (7680, 4320), (1920, 1080)):
    p, y = synth.lcg_frame(W, H, 1234)
    m = torch.zeros((W // 4) * (H // 4), dtype=torch.uint8, device="cuda")
    o = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda")
    yi, pi = api.yuv420_image(y.data_ptr(), W, H, api.CG_BT709), api.p010_image(p.data_ptr(), W, H, api.CG_BT2100)
    mi, md = api.out_image(m.data_ptr()), api.Metadata()
    g = timed(lambda: lib.uhdr_hip_generate_gainmap(C.byref(yi), C.byref(pi), api.TF_HLG, C.byref(md), C.byref(mi), 0, api.MEM_DEVICE, stream))
    mi1, oi = api.mono_image(m.data_ptr(), W // 4, H // 4), api.out_image(o.data_ptr())
    a = timed(lambda: lib.uhdr_hip_apply_gainmap(C.byref(yi), C.byref(mi1), C.byref(md), api.OUTPUT_HDR_HLG, api.FLT_MAX, C.byref(oi), api.APPLY_FAST, api.MEM_DEVICE, stream))
    q = timed(lambda: lib.uhdr_hip_apply_gainmap(C.byref(yi), C.byref(mi1), C.byref(md), api.OUTPUT_HDR_PQ, api.FLT_MAX, C.byref(oi), api.APPLY_FAST, api.MEM_DEVICE, stream))
    out["%dx%d" % (W, H)] = {"generate_us": round(g, 2), "apply_hlg_us": round(a, 2), "apply_pq_us": round(q, 2),
                             "generate_GBs": round(W * H * 4.5625 / g / 1e3, 1), "apply_GBs": round(W * H * 5.5625 / a / 1e3, 1)}
print(json.dumps(out))
