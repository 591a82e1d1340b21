#!/usr/bin/env python3
"""the bench step (generate + apply, 64 x 4K) 60 times back to back, nothing else: for kernel traces of the gaps between its kernels"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from libultrahdr_dev_amd import api
torch.cuda.set_device(0)
lib = api.init(0)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
b = bench.Batch(lib, 64, 0)
for _ in range(60):
    b.generate(stream)
    b.apply(stream, api.OUTPUT_HDR_HLG)
torch.cuda.synchronize()
