"""error of pq_oetf_est against the exact pqOetf, in 10-bit code units, over every float in [0, 64]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from libultrahdr_dev_amd import api
lib = api.init(0)
def ev(fn, x):
    out = torch.empty_like(x)
    assert lib.uhdr_hip_eval_transfer(fn, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), x.numel(), 1.0, 4.0, None) == 0
    torch.cuda.synchronize()
    return out
worst = {}
chunk = 1 << 26
for b in range(0, 0x42800000 + 1, chunk):
    n = min(chunk, 0x42800000 + 1 - b)
    x = (torch.arange(n, dtype=torch.int32, device="cuda") + b).view(torch.float32)
    a, e = ev(27, x).double() * 1023, ev(15, x).double() * 1023
    d = (a - e).abs()
    i = int(d.argmax())
    # by code range
    for lo, hi in ((0, 1), (1, 64), (64, 256), (256, 512), (512, 900), (900, 1023.5), (1023.5, 4096)):
        m = (e >= lo) & (e < hi)
        if bool(m.any()):
            worst[(lo, hi)] = max(worst.get((lo, hi), 0.0), float(d[m].max()))
    print("x from %.3g: worst %.3g at x=%.9g (code %.3f)" % (float(x[0]), float(d[i]), float(x[i]), float(e[i])), flush=True)
for k, v in sorted(worst.items()):
    print("codes [%g, %g): max |error| %.3g" % (k[0], k[1], v))
