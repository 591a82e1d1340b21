"""toneMap / convertYuv, 32 x 4K per call and one 4K frame: us per frame and GB/s of algorithmic bytes"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libultrahdr_dev_amd import api, synth
lib = api.init(0)
W, H, N = 3840, 2160, 32
# `pool` as the first argument: frames and results from a placement pool (uhdr_hip_mem_pool_*) instead of torch's allocator
if len(sys.argv) > 1 and sys.argv[1] == "pool":
    pool = api.MemPool(0, 200 << 30)
    pa, ya, da_ = pool.tensor(W * H * 3 * N), pool.tensor(W * H * 3 // 2 * N), pool.tensor(W * H * 3 // 2 * N)
    pool.trim()
    ps = [pa[i * W * H * 3:(i + 1) * W * H * 3] for i in range(N)]
    for i in range(N):
        synth.lcg_frame(W, H, 10 + i, out=(ps[i], ya[i * W * H * 3 // 2:(i + 1) * W * H * 3 // 2]))
    da_.zero_()
    ds = [da_[i * W * H * 3 // 2:(i + 1) * W * H * 3 // 2] for i in range(N)]
    print("frames and results from a placement pool")
else:
    ps = [synth.lcg_frame(W, H, 10 + i)[0] for i in range(N)]
    ds = [torch.zeros(W * H * 3 // 2, dtype=torch.uint8, device="cuda") for _ in range(N)]
    print("frames and results from torch's allocator")
sa = api.image_array([api.p010_image(p.data_ptr(), W, H, api.CG_BT2100) for p in ps])
da = api.image_array([api.yuv420_image(d.data_ptr(), W, H, api.CG_BT2100) for d in ds])
def timed(f, it=20):
    for _ in range(3): assert f() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
ms = timed(lambda: lib.uhdr_hip_tonemap_batch(N, sa, da, None))
print("toneMap     %d x 4K: %.3f ms = %.2f us per frame, %.0f GB/s" % (N, ms, ms * 1e3 / N, N * W * H * 4.5 / ms / 1e6))
ms = timed(lambda: lib.uhdr_hip_convert_yuv_batch(N, da, api.CG_BT2100, api.CG_P3, None))
print("convertYuv  %d x 4K: %.3f ms = %.2f us per frame, %.0f GB/s" % (N, ms, ms * 1e3 / N, N * W * H * 3.0 / ms / 1e6))
ms = timed(lambda: lib.uhdr_hip_tonemap_batch(1, sa, da, None), 50)
print("toneMap     one 4K: %.2f us, %.0f GB/s" % (ms * 1e3, W * H * 4.5 / ms / 1e6))
ms = timed(lambda: lib.uhdr_hip_convert_yuv_batch(1, da, api.CG_BT2100, api.CG_P3, None), 50)
print("convertYuv  one 4K: %.2f us, %.0f GB/s" % (ms * 1e3, W * H * 3.0 / ms / 1e6))
