import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from libultrahdr_dev_amd import api, synth
lib = api.init(0)
W, H = 3840, 2160
p, y = synth.lcg_frame(W, H, 1234)
m = torch.randint(0, 256, ((W // 4) * (H // 4),), dtype=torch.uint8, device="cuda")
o = torch.zeros(W * H * 8, dtype=torch.uint8, device="cuda")
yi, mi, oi = api.yuv420_image(y.data_ptr(), W, H, 0), api.mono_image(m.data_ptr(), W // 4, H // 4), api.out_image(o.data_ptr())
md = api.metadata(float(np.float32(1000.0) / np.float32(203.0)))
for mode, name in ((api.APPLY_FAST, "FAST"), (api.APPLY_EXACT, "EXACT"), (api.APPLY_EXACT_UNFILTERED, "EXACT, pre-filter off")):
    for fmt, fn in ((api.OUTPUT_HDR_HLG, "HLG"), (api.OUTPUT_HDR_PQ, "PQ"), (api.OUTPUT_HDR_LINEAR, "F16"), (api.OUTPUT_HDR_LINEAR_RGB_10BIT, "planar 10-bit")):
        f = lambda: lib.uhdr_hip_apply_gainmap(C.byref(yi), C.byref(mi), C.byref(md), fmt, api.FLT_MAX, C.byref(oi), mode, api.MEM_DEVICE, None)
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        print("%s %s: %.4f ms  %.0f MPix/s" % (name, fn, ms, W * H / 1e6 / (ms * 1e-3)))
