"""editing effects on one 4K YUV420 frame: us per call and GB/s (bytes written + the bytes they come from)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libultrahdr_dev_amd import api, synth
lib = api.init(0)
W, H = 3840, 2160
_, y = synth.lcg_frame(W, H, 1234)
out = torch.zeros(W * H * 3 // 2 + 64, dtype=torch.uint8, device="cuda")
fin = api.Image(y.data_ptr(), W, H, api.CG_BT709, None, 0, 0, api.PIX_FMT_YUV420)
fo = api.out_image(out.data_ptr())
for name, fn, fargs, obytes in (("crop 3200x1800", lib.uhdr_hip_crop, (320, 3519, 180, 1979), 3200 * 1800 * 3 // 2), ("mirror vertical", lib.uhdr_hip_mirror, (0,), W * H * 3 // 2),
                                ("mirror horizontal", lib.uhdr_hip_mirror, (1,), W * H * 3 // 2), ("rotate 90", lib.uhdr_hip_rotate, (90,), W * H * 3 // 2),
                                ("rotate 180", lib.uhdr_hip_rotate, (180,), W * H * 3 // 2), ("rotate 270", lib.uhdr_hip_rotate, (270,), W * H * 3 // 2),
                                ("resize to 1920x1080", lib.uhdr_hip_resize, (1920, 1080), 1920 * 1080 * 3 // 2), ("resize to 5760x3240", lib.uhdr_hip_resize, (5760, 3240), 5760 * 3240 * 3 // 2)):
    big = torch.zeros(max(obytes, W * H * 3 // 2) + 64, dtype=torch.uint8, device="cuda")
    fo = api.out_image(big.data_ptr())
    f = lambda: fn(C.byref(fin), *fargs, C.byref(fo), api.MEM_DEVICE, None)
    for _ in range(3): assert f() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    print("%-22s %6.1f us  %6.0f GB/s" % (name, us, 2 * obytes / us / 1e3))
