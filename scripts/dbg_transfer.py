import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from libultrahdr_dev_amd import api
lib = api.init(0)
def ev(fn, x, mn=1.0, mx=4.0):
    out = torch.empty_like(x)
    assert lib.uhdr_hip_eval_transfer(fn, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), x.numel(), mn, mx, None) == 0
    torch.cuda.synchronize()
    return out
for fn in (0, 1, 2):
    tot = 0
    for b in range(0, 0x3F800001, 1 << 27):
        n = min(1 << 27, 0x3F800001 - b)
        x = (torch.arange(n, dtype=torch.int32, device="cuda") + b).view(torch.float32)
        a, e = ev(fn, x), ev(fn + 10, x)
        bad = (a.view(torch.int32) != e.view(torch.int32))
        nb = int(bad.sum())
        tot += nb
        if nb:
            xs = x[bad][:4].cpu().numpy(); aa = a[bad][:4].cpu().numpy(); ee = e[bad][:4].cpu().numpy()
            xl = x[bad]
            print("fn", fn, "chunk", hex(b), "bad", nb, "x range", float(xl.min()), float(xl.max()))
            for i in range(len(xs)):
                print("   x=%r (0x%08x) guarded=%r exact=%r  diff_ulps=%d" % (xs[i], xs[i].view(np.uint32), aa[i], ee[i], int(aa[i].view(np.int32)) - int(ee[i].view(np.int32))))
    print("fn", fn, "total bad", tot)
