#!/usr/bin/env python3
"""Every variant of generateGainMap the reference can select (transfer function x SDR gamut x HDR gamut x sdr_is_601, ultrahdr.cpp's
luminance / gamut-conversion / YUV->RGB function pointers), as a 32 x 4K batched launch: ms per launch, TB/s of algorithmic bytes and
the fraction of 8 TB/s.  The bench's headline is ONE of these (HLG, BT.709 SDR, BT.2100 HDR); this table is what the others cost."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from libultrahdr_dev_amd import api, synth
torch.cuda.set_device(0)
lib = api.init(0)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
W, H, N = 3840, 2160, int(sys.argv[1]) if len(sys.argv) > 1 else 32


def timed(fn, iters=20):
    for _ in range(3):
        assert fn() == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


fr = [synth.lcg_frame(W, H, 4321 + i) for i in range(N)]
maps = [torch.zeros((W // 4) * (H // 4), dtype=torch.uint8, device="cuda") for _ in range(N)]
mi = api.image_array([api.out_image(m.data_ptr()) for m in maps])
mm = torch.zeros(2 * N, dtype=torch.float32, device="cuda")
GAMUT = {0: "709", 1: "P3", 2: "2100"}
TF = {api.TF_LINEAR: "LINEAR", api.TF_HLG: "HLG", api.TF_PQ: "PQ"}
px = W * H * N
worst = None
for tf in (api.TF_HLG, api.TF_PQ, api.TF_LINEAR):
    for sg in (0, 1, 2):
        for hg in (0, 1, 2):
            for is601 in (0, 1):
                yi = api.image_array([api.yuv420_image(f[1].data_ptr(), W, H, sg) for f in fr])
                pi = api.image_array([api.p010_image(f[0].data_ptr(), W, H, hg) for f in fr])
                md = api.Metadata()
                for stats in (0, 1):
                    ms = timed(lambda: lib.uhdr_hip_generate_gainmap_batch(N, yi, pi, tf, C.byref(md), mi, is601,
                                                                          C.c_void_p(mm.data_ptr()) if stats else None, s))
                    tb = px * 4.5625 / ms / 1e9
                    print("%-6s sdr %-4s hdr %-4s sdr_is_601 %d statistics %d: %.4f ms per %d frames  %.2f TB/s  %.3f of 8 TB/s" % (
                        TF[tf], GAMUT[sg], GAMUT[hg], is601, stats, ms, N, tb, tb / 8), flush=True)
                    if worst is None or tb < worst[0]:
                        worst = (tb, TF[tf], GAMUT[sg], GAMUT[hg], is601, stats)
print("slowest:", worst)
