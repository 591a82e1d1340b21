"""GPU: time uhdr_hip_jpeg_decode on one smooth 4K frame; run under rocprofv3 --kernel-trace --stats for the split."""
import ctypes as C
import os, sys, time
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
    planes = torch.zeros(W * H * 3 // 2, dtype=torch.uint8, device="cuda")
    desc = api.Image()
    for _ in range(3):
        assert lib.uhdr_hip_jpeg_decode(C.c_void_p(data.ctypes.data), data.size, C.c_void_p(planes.data_ptr()), planes.numel(), C.byref(desc), api.MEM_DEVICE, None) == 0
    t0 = time.perf_counter()
    for _ in range(10):
        lib.uhdr_hip_jpeg_decode(C.c_void_p(data.ctypes.data), data.size, C.c_void_p(planes.data_ptr()), planes.numel(), C.byref(desc), api.MEM_DEVICE, None)
    print("q%d: decode %.1f us per 4K frame (%d bytes)" % (q, (time.perf_counter() - t0) / 10 * 1e6, data.size))

# the same frame with restart intervals (libjpeg-turbo through Pillow, one interval per MCU row): every interval starts in a known state
try:
    import io
    from PIL import Image
    hy = y.cpu().numpy()
    Y = hy[:W * H].reshape(H, W)
    U = hy[W * H:W * H * 5 // 4].reshape(H // 2, W // 2)
    V = hy[W * H * 5 // 4:].reshape(H // 2, W // 2)
    ycc = np.stack([Y, np.repeat(np.repeat(U, 2, 0), 2, 1), np.repeat(np.repeat(V, 2, 0), 2, 1)], -1)
    for kw, label in ((dict(restart_marker_rows=1), "restart interval = 1 MCU row"), (dict(restart_marker_blocks=8), "restart interval = 8 MCUs"), (dict(), "no restart intervals")):
        b = io.BytesIO()
        Image.fromarray(ycc, mode="YCbCr").save(b, "JPEG", quality=95, subsampling=2, **kw)
        data = np.frombuffer(b.getvalue(), np.uint8).copy()
        for _ in range(3):
            assert lib.uhdr_hip_jpeg_decode(C.c_void_p(data.ctypes.data), data.size, C.c_void_p(planes.data_ptr()), planes.numel(), C.byref(desc), api.MEM_DEVICE, None) == 0
        t0 = time.perf_counter()
        for _ in range(10):
            lib.uhdr_hip_jpeg_decode(C.c_void_p(data.ctypes.data), data.size, C.c_void_p(planes.data_ptr()), planes.numel(), C.byref(desc), api.MEM_DEVICE, None)
        print("Pillow q95, %s: decode %.1f us per 4K frame (%d bytes)" % (label, (time.perf_counter() - t0) / 10 * 1e6, data.size))
except ImportError:
    pass
