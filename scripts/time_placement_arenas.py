#!/usr/bin/env python3
"""Is it the ALLOCATION?  Eight batches allocated one after another; per batch the step / generate / apply times, and per arena the
rate of a plain fill (torch) and of a plain read (sum) over that allocation alone."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from libultrahdr_dev_amd import api
torch.cuda.set_device(0)
lib = api.init(0)
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 8
batches = [bench.Batch(lib, 64, 0, seed_offset=65536 * r) for r in range(NB)]
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
fmt = api.OUTPUT_HDR_HLG


def ms(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def whole(frames):
    """the arena behind a list of frame views, as one int32 tensor"""
    base = frames[0]._base if frames[0]._base is not None else frames[0]
    return base[: base.numel() // 4 * 4].view(torch.int32)


for _ in range(30): batches[0].generate(s); batches[0].apply(s, fmt)
for i, b in enumerate(batches):
    g, a = ms(lambda: b.generate(s)), ms(lambda: b.apply(s, fmt))
    parts = []
    for name, fr in (("p010", b.p010), ("yuv", b.yuv), ("out", b.outs)):
        t = whole(fr)
        saved = t[:1024].clone()
        rd = ms(lambda: t.sum(), 10)
        parts.append("%s read %.0f GB/s" % (name, t.numel() * 4 / rd / 1e6))
        if name == "out":
            wr = ms(lambda: t.fill_(0), 10)
            parts.append("out fill %.0f GB/s" % (t.numel() * 4 / wr / 1e6))
    print("batch %d  generate %.4f  apply %.4f   %s" % (i, g, a, "  ".join(parts)), flush=True)
