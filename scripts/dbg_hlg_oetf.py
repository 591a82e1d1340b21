import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from libultrahdr_dev_amd import api
lib = api.init(0)
def ev(fn, x):
    out = torch.empty_like(x)
    assert lib.uhdr_hip_eval_transfer(fn, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), x.numel(), 1.0, 4.0, None) == 0
    torch.cuda.synchronize()
    return out
tot = 0
for b in range(0, 0x42800001, 1 << 27):
    n = min(1 << 27, 0x42800001 - b)
    x = (torch.arange(n, dtype=torch.int32, device="cuda") + b).view(torch.float32)
    a, e = ev(4, x), ev(14, x)
    m = a.view(torch.int32) != e.view(torch.int32)
    k = int(m.sum())
    tot += k
    if k:
        xs = x[m][:5].cpu().numpy(); print(hex(b), k, xs, a[m][:5].cpu().numpy(), e[m][:5].cpu().numpy(), x[m].min().item(), x[m].max().item())
print("total", tot)
