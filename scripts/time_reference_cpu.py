#!/usr/bin/env python3
"""The reference's own gainmapmath.cpp object code (oracle/_ref, compiled in place from /root/reference) timed beside the
oracle's restatement on one 4K LCG pair, one thread, IN THE CONTAINER (no GPU; oracle/_ref is never loaded on the GPU box).
Both must give the same bytes.  -> profiles/r02_reference_cpu.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O

W, H = 3840, 2160
O.load()
assert O.load_ref() is not None, "oracle/_ref is not built (make -C oracle ref needs /root/reference)"
p010, yuv = O.lcg_frame(W, H, 1234)
yi, pi = O.yuv420_image(yuv, W, H, O.CG_BT709), O.p010_image(p010, W, H, O.CG_BT2100)
res = {}
outs = {}
for prefix in ("orc_", "ref_"):
    t0 = time.perf_counter()
    st, gmap, md = O.generate(prefix, yi, pi, O.TF_HLG, threads=1)
    t1 = time.perf_counter()
    st2, out, _ = O.apply(prefix, yi, gmap, md, O.OUT_HDR_HLG, 3.4028234663852886e38, threads=1)
    t2 = time.perf_counter()
    assert st == 0 and st2 == 0
    outs[prefix] = (gmap, out)
    res[prefix] = {"generate_s": round(t1 - t0, 3), "apply_s": round(t2 - t1, 3), "MPix/s": round(W * H / 1e6 / (t2 - t0), 3)}
same = bool(np.array_equal(outs["orc_"][0], outs["ref_"][0]) and np.array_equal(outs["orc_"][1], outs["ref_"][1]))
doc = {"what": "one 3840x2160 LCG pair (seed 1234), generate(HLG) + apply -> RGBA1010102 HLG, 1 thread, container CPU",
       "oracle_restatement": res["orc_"], "reference_object_code": res["ref_"], "bytes_identical": same, "cores_used": 1}
assert same
json.dump(doc, open(os.path.join(ROOT, "profiles", "r02_reference_cpu.json"), "w"), indent=1)
print(json.dumps(doc))
