"""GPU: host-side timeline of one 4K decode (library built with -DUHDR_JD_TIMING prints laps to stderr)."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from libultrahdr_dev_amd import api, synth

lib = api.init(0)
W, H = 3840, 2160
_, y = synth.smooth_frame(W, H, 77)
out = torch.zeros(W * H * 2, dtype=torch.uint8, device="cuda")
n = C.c_size_t()
img = api.Image(y.data_ptr(), W, H, api.CG_BT709, y.data_ptr() + W * H, W, W // 2, api.PIX_FMT_YUV420)
for q in (95, 75):
    assert lib.uhdr_hip_jpeg_encode(C.byref(img), q, None, 0, C.c_void_p(out.data_ptr()), out.numel(), C.byref(n), api.MEM_DEVICE, None) == 0
    data = out[:n.value].cpu().numpy().copy()
    pinned = torch.from_numpy(data).pin_memory()
    planes = torch.zeros(W * H * 3 // 2, dtype=torch.uint8, device="cuda")
    desc = api.Image()
    for label, ptr in (("pageable", data.ctypes.data), ("pinned", pinned.data_ptr())):
        for it in range(4):
            sys.stderr.write("== q%d %s call %d\n" % (q, label, it)); sys.stderr.flush()
            assert lib.uhdr_hip_jpeg_decode(C.c_void_p(ptr), data.size, C.c_void_p(planes.data_ptr()), planes.numel(), C.byref(desc), api.MEM_DEVICE, None) == 0
