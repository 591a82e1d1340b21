import ctypes as C, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from libultrahdr_dev_amd import api, synth
torch.cuda.set_device(0)
lib = api.init(0)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
CASE = os.environ.get('CASE', '3840,2160,1').split(',')
for (w, h, n) in ((int(CASE[0]), int(CASE[1]), int(CASE[2])),):
    fr = [synth.lcg_frame(w, h, 1234 + i) for i in range(n)]
    maps = [torch.zeros((w // 4) * (h // 4), dtype=torch.uint8, device="cuda") for _ in range(n)]
    yi = api.image_array([api.yuv420_image(f[1].data_ptr(), w, h, api.CG_BT709) for f in fr])
    pi = api.image_array([api.p010_image(f[0].data_ptr(), w, h, api.CG_BT2100) for f in fr])
    mi = api.image_array([api.out_image(m.data_ptr()) for m in maps])
    md = api.Metadata()
    mm = torch.zeros(2 * n, dtype=torch.float32, device="cuda")
    for _ in range(20):
        rc = lib.uhdr_hip_generate_gainmap_batch(n, yi, pi, api.TF_HLG, C.byref(md), mi, 0, C.c_void_p(mm.data_ptr()), s)
        assert rc == 0
    torch.cuda.synchronize()
