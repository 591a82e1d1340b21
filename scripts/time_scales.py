"""apply of one 4K frame with gain maps of other scale factors than 4 (files from other encoders): FAST and EXACT"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from libultrahdr_dev_amd import api, synth
lib = api.init(0)
W, H = 3840, 2160
p, y = synth.lcg_frame(W, H, 1234)
o = torch.zeros(W * H * 8, dtype=torch.uint8, device="cuda")
md = api.metadata(float(np.float32(1000.0) / np.float32(203.0)))
for scale in (1, 2, 4, 8, 16):
    mw, mh = W // scale, H // scale
    m = torch.randint(0, 256, (mw * mh,), dtype=torch.uint8, device="cuda")
    yi, mi, oi = api.yuv420_image(y.data_ptr(), W, H, 0), api.mono_image(m.data_ptr(), mw, mh), api.out_image(o.data_ptr())
    for mode, name in ((api.APPLY_FAST, "FAST"), (api.APPLY_EXACT, "EXACT")):
        f = lambda: lib.uhdr_hip_apply_gainmap(C.byref(yi), C.byref(mi), C.byref(md), api.OUTPUT_HDR_HLG, api.FLT_MAX, C.byref(oi), mode, api.MEM_DEVICE, None)
        for _ in range(3): assert f() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        print("scale %2d %s -> HLG: %.1f us" % (scale, name, e0.elapsed_time(e1) / 20 * 1e3))
