"""-m gpu: the batch entry points only enqueue work -- they can be captured into a HIP graph and driven from
several host threads on their own streams (include/uhdr_hip.h: "asynchronous; graph-capturable")."""
import ctypes as C
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
FLT_MAX = 3.4028234663852886e38


def _batch(hip, orc, n, w, h, seed0):
    from tests.gpu_util import to_dev, dev_empty
    keep, yis, pis, mis, ois, host = [], [], [], [], [], []
    for i in range(n):
        p010, yuv = orc.lcg_frame(w, h, seed0 + i)
        dp, dy = to_dev(p010), to_dev(yuv)
        dm, do = dev_empty((w // 4) * (h // 4), 0), dev_empty(w * h * 4, 0)
        keep += [dp, dy]
        yis.append(hip.yuv420_image(dy.data_ptr(), w, h, hip.CG_BT709))
        pis.append(hip.p010_image(dp.data_ptr(), w, h, hip.CG_BT2100))
        mis.append(hip.out_image(dm.data_ptr()))
        ois.append(hip.out_image(do.data_ptr()))
        host.append((p010, yuv, dm, do))
    return keep, hip.image_array(yis), hip.image_array(pis), hip.image_array(mis), hip.image_array(ois), host


def _check(hip, orc, host, w, h, md):
    from tests.gpu_util import to_host, diff_1010102
    for p010, yuv, dm, do in host:
        st, omap, omd = orc.generate("orc_", orc.yuv420_image(yuv, w, h, 0), orc.p010_image(p010, w, h, 2), 1)
        assert np.array_equal(to_host(dm, omap.size).reshape(omap.shape), omap)
        st, ref, _ = orc.apply("orc_", orc.yuv420_image(yuv, w, h, 0), omap, omd, orc.OUT_HDR_HLG, FLT_MAX)
        worst, frac, ok = diff_1010102(to_host(do, w * h * 4).view(np.uint32), ref.view(np.uint32))
        assert ok and worst <= 1


def test_batch_calls_are_graph_capturable(hip, orc):
    lib = hip.load()
    n, w, h = 5, 256, 128
    keep, ya, pa, ma, oa, host = _batch(hip, orc, n, w, h, 500)
    md = hip.Metadata()
    mm = torch.zeros(2 * n, dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        # warm-up outside capture (first use may load code objects)
        s = C.c_void_p(side.cuda_stream)
        assert lib.uhdr_hip_generate_gainmap_batch(n, ya, pa, hip.TF_HLG, C.byref(md), ma, 0, C.c_void_p(mm.data_ptr()), s) == 0
        assert lib.uhdr_hip_apply_gainmap_batch(n, ya, ma, C.byref(md), hip.OUTPUT_HDR_HLG, FLT_MAX, oa, hip.APPLY_FAST, s) == 0
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for _, _, dm, do in host:
        dm.zero_(); do.zero_()
    mm.zero_()
    with torch.cuda.graph(g, stream=side):
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        assert lib.uhdr_hip_generate_gainmap_batch(n, ya, pa, hip.TF_HLG, C.byref(md), ma, 0, C.c_void_p(mm.data_ptr()), s) == 0
        assert lib.uhdr_hip_apply_gainmap_batch(n, ya, ma, C.byref(md), hip.OUTPUT_HDR_HLG, FLT_MAX, oa, hip.APPLY_FAST, s) == 0
    torch.cuda.synchronize()
    assert all(int(dm.sum()) == 0 for _, _, dm, _ in host), "capture must not execute the kernels"
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    _check(hip, orc, host, w, h, md)
    assert float(mm[0]) <= float(mm[1])


def test_two_host_threads_two_streams(hip, orc):
    lib = hip.load()
    w, h, n = 192, 96, 6
    sets = [_batch(hip, orc, n, w, h, 700 + 50 * t) for t in range(2)]
    errs = []

    def worker(t):
        try:
            torch.cuda.set_device(0)
            st = torch.cuda.Stream()
            keep, ya, pa, ma, oa, host = sets[t]
            md = hip.Metadata()
            s = C.c_void_p(st.cuda_stream)
            for _ in range(5):
                assert lib.uhdr_hip_generate_gainmap_batch(n, ya, pa, hip.TF_HLG, C.byref(md), ma, 0, None, s) == 0
                assert lib.uhdr_hip_apply_gainmap_batch(n, ya, ma, C.byref(md), hip.OUTPUT_HDR_HLG, FLT_MAX, oa, hip.APPLY_FAST, s) == 0
            st.synchronize()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    torch.cuda.synchronize()
    md = hip.metadata(float(np.float32(1000.0) / np.float32(203.0)))
    for t in range(2):
        _check(hip, orc, sets[t][5], w, h, md)
