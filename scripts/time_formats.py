"""FAST apply of a 64-frame 4K batch, every output format: ms per launch, us per frame, GB/s of algorithmic bytes."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from libultrahdr_dev_amd import api, synth
lib = api.init(0)
W, H, N = 3840, 2160, int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames = [synth.lcg_frame(W, H, 100 + i)[1] for i in range(N)]
maps = [torch.randint(0, 256, ((W // 4) * (H // 4),), dtype=torch.uint8, device="cuda") for _ in range(N)]
outs = [torch.zeros(W * H * 8, dtype=torch.uint8, device="cuda") for _ in range(N)]
ya = api.image_array([api.yuv420_image(f.data_ptr(), W, H, 0) for f in frames])
ma = api.image_array([api.mono_image(m.data_ptr(), W // 4, H // 4) for m in maps])
oa = api.image_array([api.out_image(o.data_ptr()) for o in outs])
md = api.metadata(float(np.float32(1000.0) / np.float32(203.0)))
for boost, bn in ((api.FLT_MAX, "display boost = content boost"), (2.0, "display boost 2 < content boost")):
    for fmt, fn, bpp in ((api.OUTPUT_HDR_HLG, "HLG", 4), (api.OUTPUT_HDR_PQ, "PQ", 4), (api.OUTPUT_HDR_LINEAR, "F16", 8), (api.OUTPUT_HDR_LINEAR_RGB_10BIT, "planar 10-bit", 6)):
        f = lambda: lib.uhdr_hip_apply_gainmap_batch(N, ya, ma, C.byref(md), fmt, boost, oa, api.APPLY_FAST, None)
        for _ in range(3): assert f() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        byts = N * (W * H * 1.5 + W * H / 16 + W * H * bpp)
        print("%-32s %-14s %.3f ms / %d frames = %.1f us per frame, %.0f GB/s" % (bn, fn, ms, N, ms * 1e3 / N, byts / (ms * 1e-3) / 1e9))
