#!/usr/bin/env python3
"""one 4K apply -> HLG, FAST, 200 launches (for kernel traces)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from libultrahdr_dev_amd import api, synth
lib = api.init(0)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
p, y = synth.lcg_frame(W, H, 1234)
m = torch.randint(0, 255, ((W // 4) * (H // 4),), dtype=torch.uint8, device="cuda")
o = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda")
yi = api.yuv420_image(y.data_ptr(), W, H, api.CG_BT709)
md = api.Metadata()
pi, mi = api.p010_image(p.data_ptr(), W, H, api.CG_BT2100), api.out_image(m.data_ptr())
assert lib.uhdr_hip_generate_gainmap(C.byref(yi), C.byref(pi), api.TF_HLG, C.byref(md), C.byref(mi), 0, api.MEM_DEVICE, stream) == 0
mi1, oi = api.mono_image(m.data_ptr(), W // 4, H // 4), api.out_image(o.data_ptr())
for _ in range(200):
    rc = lib.uhdr_hip_apply_gainmap(C.byref(yi), C.byref(mi1), C.byref(md), api.OUTPUT_HDR_HLG, api.FLT_MAX, C.byref(oi), api.APPLY_FAST, api.MEM_DEVICE, stream)
    assert rc == 0, rc
torch.cuda.synchronize()
