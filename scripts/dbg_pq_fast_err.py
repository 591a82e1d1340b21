"""GPU: relative error of pq_inv_oetf_fast (eval code 22) against the exact f64 form (12) for every float in (1e-4, 1], by octave"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from libultrahdr_dev_amd import api
lib = api.init(0)
def ev(fn, x):
    out = torch.empty_like(x)
    assert lib.uhdr_hip_eval_transfer(fn, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), x.numel(), 1.0, 4.0, None) == 0
    return out
bits = lambda v: int(np.float32(v).view(np.uint32))
lo = bits(np.nextafter(np.float32(1e-4), np.float32(1)))
worst_all = 0.0
b = lo
while b <= 0x3F800000:
    n = min(1 << 23, 0x3F800000 + 1 - b)
    x = (torch.arange(n, dtype=torch.int32, device="cuda") + b).view(torch.float32)
    a, e = ev(22, x).double(), ev(12, x).double()
    rel = ((a - e).abs() / e)
    w = float(rel.max()); i = int(rel.argmax())
    print("[%.3e, %.3e]: worst rel %.3e at %.9g (fast %.9g exact %.9g)" % (float(x[0]), float(x[-1]), w, float(x[i]), float(a[i]), float(e[i])))
    worst_all = max(worst_all, w)
    b += n
z = torch.tensor([0.0, 1e-4, 5e-5], device="cuda")
print("zeros:", ev(22, z).tolist(), ev(12, z).tolist(), "worst overall %.3e" % worst_all)
