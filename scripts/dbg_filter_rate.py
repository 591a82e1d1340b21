"""GPU, instrumentation build (-DUHDR_GEN_COUNT): how often k_generate's waves leave the f32 fast path."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libultrahdr_dev_amd import api, synth

lib = api.init(0)
W, H, N = 3840, 2160, 16
fr = [synth.lcg_frame(W, H, 1234 + i) for i in range(N)]
maps = [torch.zeros((W // 4) * (H // 4), dtype=torch.uint8, device="cuda") for _ in range(N)]
ya = api.image_array([api.yuv420_image(f[1].data_ptr(), W, H, api.CG_BT709) for f in fr])
pa = api.image_array([api.p010_image(f[0].data_ptr(), W, H, api.CG_BT2100) for f in fr])
ma = api.image_array([api.out_image(m.data_ptr()) for m in maps])
mm = torch.zeros(2 * N, dtype=torch.float32, device="cuda")
md = api.Metadata()
raw = C.CDLL(api.LIB_PATH)
cnt = (C.c_ulonglong * 4)()
for label, stats in (("no statistics", None), ("with statistics", C.c_void_p(mm.data_ptr()))):
    raw.uhdr_hip_debug_counters(cnt, 1)
    assert lib.uhdr_hip_generate_gainmap_batch(N, ya, pa, api.TF_HLG, C.byref(md), ma, 0, stats, None) == 0
    torch.cuda.synchronize()
    raw.uhdr_hip_debug_counters(cnt, 1)
    wt, ex, sp, dp = [int(v) for v in cnt]
    print("%s: wave-tiles %d, on exact path %d (%.2f%%), exact statistics passes %d (%.2f%% of waves), doubtful pixels %d (%.4f%% of pixels)"
          % (label, wt, ex, 100.0 * ex / max(wt, 1), sp, 100.0 * sp / max(wt / 4, 1), dp, 100.0 * dp / max(wt * 128, 1)))
