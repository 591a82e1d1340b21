import ctypes as C, os, sys
sys.path.insert(0, '/root/repo')
import torch, bench
from libultrahdr_dev_amd import api
torch.cuda.set_device(0)
lib = api.init(0)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
b = bench.Batch(lib, 64, 0)
for it in range(2):
    b.generate(stream); torch.cuda.synchronize()
    mm = b.minmax.cpu().view(-1, 2)
    print("iter", it, "candidates per image", mm[:8, 0].tolist(), "entries", mm[:8, 1].tolist(), "mean cand %.1f entries %.1f max cand %.0f" % (mm[:, 0].mean(), mm[:, 1].mean(), mm[:, 0].max()))
