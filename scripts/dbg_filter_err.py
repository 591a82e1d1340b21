"""GPU: worst relative error of the f32 fast transfer functions against the exact ones, every float of the domain."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libultrahdr_dev_amd import api

lib = api.init(0)

def ev(fn, x):
    out = torch.empty_like(x)
    assert lib.uhdr_hip_eval_transfer(fn, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), x.numel(), 1.0, 4.0, None) == 0
    torch.cuda.synchronize()
    return out

def scan(fast, exact, lo_bits, hi_bits, name):
    worst = 0.0; worst_x = 0.0
    chunk = 1 << 27
    for b in range(lo_bits, hi_bits + 1, chunk):
        n = min(chunk, hi_bits + 1 - b)
        x = (torch.arange(n, dtype=torch.int32, device="cuda") + b).view(torch.float32)
        a, e = ev(fast, x).double(), ev(exact, x).double()
        rel = ((a - e).abs() / e.abs().clamp_min(1e-300))
        rel[e == 0] = (a[e == 0] != 0).double()
        m, i = rel.max(0)
        if float(m) > worst:
            worst, worst_x = float(m), float(x[i])
    print("%s: worst rel err %.3e at x=%.9g" % (name, worst, worst_x))

import numpy as np
lo = int(np.float32(1e-12).view(np.uint32))
scan(20, 10, lo, 0x3F800000, "srgb_inv fast vs exact, [1e-12,1]")
scan(21, 11, lo, 0x3F800000, "hlg_inv fast vs exact, [1e-12,1]")
scan(20, 10, int(np.float32(0.04045).view(np.uint32)) + 1, 0x3F800000, "srgb_inv fast vs exact, (0.04045,1]")
scan(21, 11, int(np.float32(0.5).view(np.uint32)) + 1, 0x3F800000, "hlg_inv fast vs exact, (0.5,1]")
# log2: absolute error vs double log2 on [0.25, 64]
worst = 0.0
for b in range(0x3E800000, 0x42800000 + 1, 1 << 27):
    n = min(1 << 27, 0x42800000 + 1 - b)
    x = (torch.arange(n, dtype=torch.int32, device="cuda") + b).view(torch.float32)
    a = ev(23, x).double()
    worst = max(worst, float((a - torch.log2(x.double())).abs().max()))
print("v_log_f32 abs err on [0.25,64]: %.3e" % worst)
