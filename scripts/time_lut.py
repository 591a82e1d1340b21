"""LUT-mode apply: one 4K frame and a 32-frame launch, every output format"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from libultrahdr_dev_amd import api, synth
lib = api.init(0)
W, H, N = 3840, 2160, 32
frames = [synth.lcg_frame(W, H, 100 + i)[1] for i in range(N)]
maps = [torch.randint(0, 256, ((W // 4) * (H // 4),), dtype=torch.uint8, device="cuda") for _ in range(N)]
outs = [torch.zeros(W * H * 8, dtype=torch.uint8, device="cuda") for _ in range(N)]
ya = api.image_array([api.yuv420_image(f.data_ptr(), W, H, 0) for f in frames])
ma = api.image_array([api.mono_image(m.data_ptr(), W // 4, H // 4) for m in maps])
oa = api.image_array([api.out_image(o.data_ptr()) for o in outs])
md = api.metadata(float(np.float32(1000.0) / np.float32(203.0)))
for n in (1, N):
    for fmt, fn in ((api.OUTPUT_HDR_HLG, "HLG"), (api.OUTPUT_HDR_PQ, "PQ"), (api.OUTPUT_HDR_LINEAR, "F16"), (api.OUTPUT_HDR_LINEAR_RGB_10BIT, "planar 10-bit")):
        f = lambda: lib.uhdr_hip_apply_gainmap_batch(n, ya, ma, C.byref(md), fmt, api.FLT_MAX, oa, api.APPLY_LUT, None)
        for _ in range(3): assert f() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("LUT mode, %2d frame(s), %-14s %.3f ms = %.1f us per frame" % (n, fn, ms, ms * 1e3 / n))
