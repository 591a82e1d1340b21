#!/usr/bin/env python3
"""the bench step (generate + apply, 64 x 4K) issued on ONE stream, and alternately on TWO streams with maps / outputs of their own
(consecutive steps are independent batches: what a service with a queue of batches would do): ms per step, whole job.
Round 3, four boxes: the second stream hides k_generate_resolve's latency on some runs (0.990 -> 0.963 ms, three repeats alike) and
loses on others (0.973 -> 0.986; inside bench.py, behind the timed region: 0.964 -> 1.08): not a dependable gain, so bench.py's
`value` and the library stay with one stream per caller and leave the overlap to callers that have independent batches."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from libultrahdr_dev_amd import api
torch.cuda.set_device(0)
lib = api.init(0)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
b0 = bench.Batch(lib, frames, 0)
b1 = bench.Batch(lib, frames, 0)          # (frames of its own as well: 64 x 37 MB more)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
hs = [C.c_void_p(s.cuda_stream) for s in streams]
fmt = api.OUTPUT_HDR_HLG


def run(nstreams, steps):
    bs = [b0, b1]
    for k in range(10):
        i = k % nstreams
        with torch.cuda.stream(streams[i]):
            bs[i].generate(hs[i]); bs[i].apply(hs[i], fmt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        i = k % nstreams
        with torch.cuda.stream(streams[i]):
            bs[i].generate(hs[i]); bs[i].apply(hs[i], fmt)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for _ in range(200):   # clocks
    b0.generate(hs[0]); b0.apply(hs[0], fmt)
torch.cuda.synchronize()
for rep in range(3):
    for ns in (1, 2):
        ms = run(ns, int(os.environ.get('STEPS', '100')))
        print("streams %d: %.4f ms per step, %.0f MPix/s" % (ns, ms, frames * 3840 * 2160 / ms / 1e3))
