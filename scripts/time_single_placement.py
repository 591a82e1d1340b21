#!/usr/bin/env python3
"""Single-image calls (BASELINE configs[1] / configs[4]) on buffers from torch's allocator against buffers from a placement pool
(uhdr_hip_mem_pool_*): one 8K apply -> PQ / F16, one 4K generate, one 4K apply.  Several allocations of each kind, all kept."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from libultrahdr_dev_amd import api, synth
torch.cuda.set_device(0)
lib = api.init(0)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(f, n=40):
    for _ in range(5): assert f() == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        for _ in range(10): f()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def case(alloc, label):
    w8, h8 = 7680, 4320
    y8, m8, o8 = alloc(w8 * h8 * 3 // 2), alloc((w8 // 4) * (h8 // 4)), alloc(w8 * h8 * 8)
    p8 = alloc(w8 * h8 * 3)
    synth.lcg_frame(w8, h8, 1234, out=(p8, y8))
    m8.copy_(torch.randint(0, 256, (m8.numel(),), dtype=torch.uint8, device="cuda"))
    yi8, mi8, oi8 = api.yuv420_image(y8.data_ptr(), w8, h8, api.CG_BT709), api.mono_image(m8.data_ptr(), w8 // 4, h8 // 4), api.out_image(o8.data_ptr())
    md8 = api.metadata(float(np.float32(10000.0) / np.float32(203.0)))
    pq = timed(lambda: lib.uhdr_hip_apply_gainmap(C.byref(yi8), C.byref(mi8), C.byref(md8), api.OUTPUT_HDR_PQ, api.FLT_MAX, C.byref(oi8), api.APPLY_FAST, api.MEM_DEVICE, s))
    f16 = timed(lambda: lib.uhdr_hip_apply_gainmap(C.byref(yi8), C.byref(mi8), C.byref(md8), api.OUTPUT_HDR_LINEAR, api.FLT_MAX, C.byref(oi8), api.APPLY_FAST, api.MEM_DEVICE, s))
    w, h = 3840, 2160
    p4, y4, m4, o4 = alloc(w * h * 3), alloc(w * h * 3 // 2), alloc((w // 4) * (h // 4)), alloc(w * h * 4)
    synth.lcg_frame(w, h, 1234, out=(p4, y4))
    yi, pi, mo, oo = api.yuv420_image(y4.data_ptr(), w, h, api.CG_BT709), api.p010_image(p4.data_ptr(), w, h, api.CG_BT2100), api.out_image(m4.data_ptr()), api.out_image(o4.data_ptr())
    md = api.Metadata()
    g = timed(lambda: lib.uhdr_hip_generate_gainmap(C.byref(yi), C.byref(pi), api.TF_HLG, C.byref(md), C.byref(mo), 0, api.MEM_DEVICE, s))
    mi = api.mono_image(m4.data_ptr(), w // 4, h // 4)
    a = timed(lambda: lib.uhdr_hip_apply_gainmap(C.byref(yi), C.byref(mi), C.byref(md), api.OUTPUT_HDR_HLG, api.FLT_MAX, C.byref(oo), api.APPLY_FAST, api.MEM_DEVICE, s))
    print("%-34s 8K apply -> PQ %.1f us  -> F16 %.1f us   4K generate %.1f us  4K apply -> HLG %.1f us" % (label, pq, f16, g, a), flush=True)
    return (y8, m8, o8, p8, p4, y4, m4, o4)


keep = []
for k in range(4):
    keep.append(case(lambda n: torch.empty(n, dtype=torch.uint8, device="cuda"), "torch allocator, set %d" % k))
for chunk in (2 << 20, 16 << 20):
    for k in range(3):
        pool = api.MemPool(0, 3 << 30, chunk)
        keep.append(pool)
        keep.append(case(pool.tensor, "pool of 3 GiB, %d MiB chunks, set %d" % (chunk >> 20, k)))
