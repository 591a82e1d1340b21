"""UHDR_HIP_MEM_HOST generate + apply of 4K pairs from 1, 2 and 4 host threads (each on its own stream): pairs per second.
The box's host link moves ~55 GB/s in both directions together (scripts/pcie_rates.py), 84 MB per pair: ~650 pairs/s at most."""
import sys, os, time, threading, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from libultrahdr_dev_amd import api, synth
lib = api.init(0)
W, H = 3840, 2160
FLT_MAX = api.FLT_MAX
p, y = synth.lcg_frame(W, H, 1)
hp, hy = p.cpu().numpy(), y.cpu().numpy()

def worker(n, out, t):
    torch.cuda.set_device(0)
    st = torch.cuda.Stream()
    s = C.c_void_p(st.cuda_stream)
    gmap = np.zeros((W // 4) * (H // 4), np.uint8)
    hout = np.zeros(W * H, np.uint32)
    yi, pi = api.yuv420_image(hy.ctypes.data, W, H, api.CG_BT709), api.p010_image(hp.ctypes.data, W, H, api.CG_BT2100)
    mi, oi = api.out_image(gmap.ctypes.data), api.out_image(hout.ctypes.data)
    mm = api.mono_image(gmap.ctypes.data, W // 4, H // 4)
    md = api.Metadata()
    for _ in range(n):
        assert lib.uhdr_hip_generate_gainmap(C.byref(yi), C.byref(pi), api.TF_HLG, C.byref(md), C.byref(mi), 0, api.MEM_HOST, s) == 0
        assert lib.uhdr_hip_apply_gainmap(C.byref(yi), C.byref(mm), C.byref(md), api.OUTPUT_HDR_HLG, FLT_MAX, C.byref(oi), api.APPLY_FAST, api.MEM_HOST, s) == 0
    out[t] = int(hout[:1000].sum())

for threads in (1, 2, 4, 8, 12):
    out = {}
    worker(3, out, -1)
    th = [threading.Thread(target=worker, args=(40, out, t)) for t in range(threads)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print("%d host thread(s): %.0f pairs/s (%.3f ms per pair per thread, %.1f GB/s over the link)" % (threads, threads * 40 / dt, dt / 40 * 1e3, threads * 40 * 84.0e6 / dt / 1e9), flush=True)
