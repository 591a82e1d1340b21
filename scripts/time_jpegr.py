"""GPU: time the whole-codec entry points on one smooth 4K frame pair (device memory in, file on the host) and on BASELINE configs[0]
(640x480): uhdr_hip_jpegr_encode_api1 / api0, uhdr_hip_jpegr_decode.  Run under rocprofv3 --kernel-trace --stats for the split."""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from libultrahdr_dev_amd import api, synth

lib = api.init(0)
FLT_MAX = 3.4028234663852886e38
for W, H in ((3840, 2160), (640, 480)):
    p, y = synth.smooth_frame(W, H, 77)
    pi = api.p010_image(p.data_ptr(), W, H, api.CG_BT2100)
    yi = api.yuv420_image(y.data_ptr(), W, H, api.CG_BT709)
    out = np.zeros(W * H * 3, np.uint8)
    n = C.c_size_t()

    def enc1():
        return lib.uhdr_hip_jpegr_encode_api1(C.byref(pi), C.byref(yi), api.TF_HLG, 95, None, 0, C.c_void_p(out.ctypes.data), out.size, C.byref(n), api.MEM_DEVICE, None)

    def enc0():
        return lib.uhdr_hip_jpegr_encode_api0(C.byref(pi), api.TF_HLG, 95, None, 0, C.c_void_p(out.ctypes.data), out.size, C.byref(n), api.MEM_DEVICE, None)

    def timed(fn, iters=20):
        for _ in range(3):
            assert fn() == 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    t0 = timed(enc0)
    t1 = timed(enc1)
    data = out[:n.value].copy()
    dst = torch.zeros(W * H * 8, dtype=torch.uint8, device="cuda")
    dest, md = api.Image(), api.Metadata()

    def dec(fmt=api.OUTPUT_HDR_HLG, mode=api.APPLY_FAST):
        return lib.uhdr_hip_jpegr_decode(C.c_void_p(data.ctypes.data), data.size, fmt, FLT_MAX, C.c_void_p(dst.data_ptr()), dst.numel(), C.byref(dest), C.byref(md), mode,
                                         api.MEM_DEVICE, None)
    td = timed(dec)
    tdl = timed(lambda: dec(api.OUTPUT_HDR_LINEAR))
    print("%dx%d: encode API-0 %.3f ms, API-1 %.3f ms (%d bytes, q95); decode -> HLG 1010102 %.3f ms, -> linear F16 %.3f ms" % (W, H, t0, t1, data.size, td, tdl), flush=True)

# n files per call (uhdr_hip_jpegr_decode_batch): the latency-bound JPEG decodes overlap
W, H = 3840, 2160
for nfiles in (1, 2, 4, 8, 16):
    blobs = []
    for i in range(nfiles):
        p, y = synth.smooth_frame(W, H, 100 + i)
        pi = api.p010_image(p.data_ptr(), W, H, api.CG_BT2100)
        yi = api.yuv420_image(y.data_ptr(), W, H, api.CG_BT709)
        out = np.zeros(W * H * 3, np.uint8)
        n = C.c_size_t()
        assert lib.uhdr_hip_jpegr_encode_api1(C.byref(pi), C.byref(yi), api.TF_HLG, 95, None, 0, C.c_void_p(out.ctypes.data), out.size, C.byref(n), api.MEM_DEVICE, None) == 0
        blobs.append(out[:n.value].copy())
    ptrs = (C.c_void_p * nfiles)(*[b.ctypes.data for b in blobs])
    sizes = (C.c_size_t * nfiles)(*[b.size for b in blobs])
    outs = [torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda") for _ in range(nfiles)]
    optr = (C.c_void_p * nfiles)(*[o.data_ptr() for o in outs])
    ocap = (C.c_size_t * nfiles)(*[o.numel() for o in outs])
    dests, mds, status = (api.Image * nfiles)(), (api.Metadata * nfiles)(), (C.c_int * nfiles)()
    call = lambda: lib.uhdr_hip_jpegr_decode_batch(nfiles, ptrs, sizes, api.OUTPUT_HDR_HLG, FLT_MAX, optr, ocap, dests, mds, status, api.APPLY_FAST, api.MEM_DEVICE, None)
    for _ in range(3):
        assert call() == 0
    t0 = time.perf_counter()
    for _ in range(10):
        call()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print("decode batch of %2d 4K files: %.3f ms per call, %.3f ms per file, %.0f MPix/s" % (nfiles, ms, ms / nfiles, nfiles * W * H / 1e6 / (ms * 1e-3)), flush=True)
