#!/usr/bin/env python3
"""the bench step (generate + apply, 64 x 4K) issued eagerly and replayed from a captured hipGraph (torch.cuda.CUDAGraph): ms per step.
Round 3, one box: eager 0.999-1.005 ms, graph of one step 1.007-1.009, graph of five steps 1.003-1.008: the step is not launch-bound."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from libultrahdr_dev_amd import api
torch.cuda.set_device(0)
lib = api.init(0)
b = bench.Batch(lib, 64, 0)
fmt = api.OUTPUT_HDR_HLG
side = torch.cuda.Stream()
hs = C.c_void_p(side.cuda_stream)
with torch.cuda.stream(side):
    for _ in range(3):
        b.generate(hs); b.apply(hs, fmt)
side.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    b.generate(C.c_void_p(torch.cuda.current_stream().cuda_stream)); b.apply(C.c_void_p(torch.cuda.current_stream().cuda_stream), fmt)
g5 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g5, stream=side):
    for _ in range(5):
        b.generate(C.c_void_p(torch.cuda.current_stream().cuda_stream)); b.apply(C.c_void_p(torch.cuda.current_stream().cuda_stream), fmt)
s0 = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(300):
    b.generate(s0); b.apply(s0, fmt)
torch.cuda.synchronize()


def t(fn, steps_per_call, calls):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(calls):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (calls * steps_per_call) * 1e3


def eager():
    b.generate(s0); b.apply(s0, fmt)


for rep in range(3):
    print("eager %.4f ms | graph of 1 step %.4f | graph of 5 steps %.4f" % (t(eager, 1, 40), t(g.replay, 1, 40), t(g5.replay, 5, 8)))
