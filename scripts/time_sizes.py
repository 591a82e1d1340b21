#!/usr/bin/env python3
"""batched generate + apply (FAST -> HLG 1010102) for other frame sizes than the bench's 64 x 4K: ms per launch and TB/s of algorithmic bytes"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from libultrahdr_dev_amd import api, synth
torch.cuda.set_device(0)
lib = api.init(0)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:   # (clocks: the card raises them only under load)
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


CASES = ((3840, 2160, 64), (1920, 1080, 64), (1280, 720, 64), (640, 480, 64), (7680, 4320, 16), (3840, 2160, 8), (3840, 2160, 4), (3840, 2160, 2), (3840, 2160, 1), (1920, 1080, 8), (1920, 1080, 2), (4000, 3000, 32), (1000, 752, 64))
for (w, h, n) in CASES:
    fr = [synth.lcg_frame(w, h, 1234 + i) for i in range(n)]
    maps = [torch.zeros(((w + 3) // 4) * ((h + 3) // 4), dtype=torch.uint8, device="cuda") for _ in range(n)]
    outs = [torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda") for _ in range(n)]
    yi = api.image_array([api.yuv420_image(f[1].data_ptr(), w, h, api.CG_BT709) for f in fr])
    pi = api.image_array([api.p010_image(f[0].data_ptr(), w, h, api.CG_BT2100) for f in fr])
    mi = api.image_array([api.out_image(m.data_ptr()) for m in maps])
    oi = api.image_array([api.out_image(o.data_ptr()) for o in outs])
    md = api.Metadata()
    mm = torch.zeros(2 * n, dtype=torch.float32, device="cuda")
    g = timed(lambda: lib.uhdr_hip_generate_gainmap_batch(n, yi, pi, api.TF_HLG, C.byref(md), mi, 0, C.c_void_p(mm.data_ptr()), s))
    a = timed(lambda: lib.uhdr_hip_apply_gainmap_batch(n, yi, mi, C.byref(md), api.OUTPUT_HDR_HLG, api.FLT_MAX, oi, api.APPLY_FAST, s))
    px = w * h * n
    print("%5dx%-5d x %2d: generate %.4f ms (%.2f TB/s)  apply %.4f ms (%.2f TB/s)  step %.0f MPix/s" % (
        w, h, n, g, px * 4.5625 / g / 1e9, a, px * 5.5625 / a / 1e9, px / (g + a) / 1e3))
    del fr, maps, outs
    torch.cuda.empty_cache()
