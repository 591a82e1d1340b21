"""-m gpu: randomized parity sweep in the spirit of the reference's fuzzers (fuzzer/ultrahdr_enc_fuzzer.cpp:87-319:
random even dimensions, gamuts, transfer function, strides, separate chroma planes): every draw runs generate ->
apply (random output format, display boost, EXACT or FAST; two draws in three also through the opt-in LUT pipelines) -> toneMap ->
convertYuv on the GPU and on the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
FLT_MAX = 3.4028234663852886e38


def _planes(rng, w, h, ls, ycs, pcs, kind):
    if kind == 0:
        y8 = rng.randint(0, 256, (h, w)); u8 = rng.randint(0, 256, (h // 2, w // 2)); v8 = rng.randint(0, 256, (h // 2, w // 2))
        py = rng.randint(64, 941, (h, w)) << 6; pc = rng.randint(64, 961, (h // 2, w)) << 6
    else:  # flat / saturated regions: clamps, gain == min / max, zero luminance
        base = rng.randint(0, 256)
        y8 = np.full((h, w), base); u8 = np.full((h // 2, w // 2), rng.randint(0, 256)); v8 = np.full((h // 2, w // 2), rng.randint(0, 256))
        py = np.full((h, w), rng.choice([64, 940, 502, 0, 1023]) << 6); pc = np.full((h // 2, w), rng.choice([64, 960, 512]) << 6)
        y8[: h // 2, : w // 2] = rng.choice([0, 255]); py[h // 2:, :] = rng.choice([64, 940]) << 6
    yl = np.zeros((h, ls), np.uint8); yl[:, :w] = y8
    yc = np.zeros((h, ycs), np.uint8); yc[: h // 2, : w // 2] = u8; yc[h // 2: 2 * (h // 2), : w // 2] = v8
    pl = np.zeros((h, ls), np.uint16); pl[:, :w] = py
    pcc = np.zeros((h // 2, pcs), np.uint16); pcc[:, :w] = pc
    return yl.reshape(-1), yc.reshape(-1), pl.reshape(-1), pcc.reshape(-1)


@pytest.mark.parametrize("seed", range(int(os.environ.get("UHDR_FUZZ_SEEDS", "24"))))   # scripts/long_checks.sh runs 1000
def test_random_configuration(hip, orc, seed):
    from tests.gpu_util import to_dev, dev_empty, to_host, stream_ptr, diff_1010102, half_ulp_diff
    lib, L = hip.load(), orc.load()
    rng = np.random.RandomState(1000 + seed)
    w, h = int(rng.randint(4, 130)) * 2, int(rng.randint(4, 70)) * 2
    if seed % 3 == 0:
        w, h = w - w % 8, h - h % 4          # make the vector paths likely too
        w, h = max(w, 8), max(h, 8)
    pad = int(rng.choice([0, 0, 8, 5]))
    ls, ycs, pcs = w + pad, w // 2 + int(rng.choice([0, 4, 3])), w + int(rng.choice([0, 8, 2]))
    sg, hg, tf, is601 = int(rng.randint(0, 3)), int(rng.randint(0, 3)), int(rng.randint(0, 3)), bool(rng.randint(0, 2))
    yl, yc, pl, pc = _planes(rng, w, h, ls, ycs, pcs, seed % 2)
    off = int(rng.choice([0, 0, 1, 3]))
    d_yl, d_yc = to_dev(np.concatenate([np.zeros(off, np.uint8), yl])), to_dev(np.concatenate([np.zeros(off, np.uint8), yc]))
    d_pl, d_pc = to_dev(np.concatenate([np.zeros(off, np.uint16), pl])), to_dev(np.concatenate([np.zeros(off, np.uint16), pc]))
    yi = hip.yuv420_image(d_yl.data_ptr() + off, w, h, sg, ls, ycs, d_yc.data_ptr() + off)
    pi = hip.p010_image(d_pl.data_ptr() + 2 * off, w, h, hg, ls, pcs, d_pc.data_ptr() + 2 * off)
    oyi, opi = orc.yuv420_image(yl, w, h, sg, ls, ycs, yc), orc.p010_image(pl, w, h, hg, ls, pcs, pc)
    s = stream_ptr()
    # generate
    mw, mh = w // 4, h // 4
    dmap = dev_empty(mw * mh, 0xCD)
    dest, md = hip.out_image(dmap.data_ptr()), hip.Metadata()
    assert lib.uhdr_hip_generate_gainmap(C.byref(yi), C.byref(pi), tf, C.byref(md), C.byref(dest), int(is601), hip.MEM_DEVICE, s) == 0
    st, omap, omd = orc.generate("orc_", oyi, opi, tf, is601)
    gmap = to_host(dmap, mw * mh).reshape(mh, mw)
    assert st == 0 and np.array_equal(gmap, omap), (w, h, sg, hg, tf, is601, int((gmap != omap).sum()))
    if seed % 3 == 1:   # the opt-in LUT pipeline on the same draw (ultrahdr.cpp's USE_*_LUT branches: DESIGN 4.3)
        lmap = dev_empty(mw * mh, 0xCD)
        ldest, lmd = hip.out_image(lmap.data_ptr()), hip.Metadata()
        assert lib.uhdr_hip_generate_gainmap_ex(C.byref(yi), C.byref(pi), tf, C.byref(lmd), C.byref(ldest), int(is601), hip.GENERATE_LUT,
                                                hip.MEM_DEVICE, s) == 0
        lst, lomap, _ = orc.generate("orc_", oyi, opi, tf, is601, lut=True)
        lg = to_host(lmap, mw * mh).reshape(mh, mw)
        assert lst == 0 and np.array_equal(lg, lomap), ("LUT generate", w, h, sg, hg, tf, is601, int((lg != lomap).sum()))
    # apply on a map whose size divides the image (random integer scale)
    scale = int(rng.choice([1, 2, 4, 4, 4]))
    aw, ah = (w // (2 * scale)) * 2 * scale, (h // (2 * scale)) * 2 * scale
    if aw >= 2 * scale and ah >= 2 * scale:
        amap = rng.randint(0, 256, (ah // scale, aw // scale)).astype(np.uint8)
        fmt = int(rng.choice([1, 2, 3, 4]))
        boost = float(rng.choice([FLT_MAX, 2.0, omd.maxContentBoost]))
        mode = hip.APPLY_EXACT if seed % 4 == 0 else hip.APPLY_FAST
        ayi = hip.yuv420_image(d_yl.data_ptr() + off, aw, ah, sg, ls, ycs, d_yc.data_ptr() + off)
        # the V plane offset depends on the image height (gainmapmath.cpp:568): re-describe for the cropped height
        oayi = orc.yuv420_image(yl, aw, ah, sg, ls, ycs, yc)
        dam = to_dev(amap)
        dout = dev_empty(hip.output_bytes(fmt, aw, ah), 0xCD)
        mi, od = hip.mono_image(dam.data_ptr(), aw // scale, ah // scale), hip.out_image(dout.data_ptr())
        amd = hip.metadata(omd.maxContentBoost)
        assert lib.uhdr_hip_apply_gainmap(C.byref(ayi), C.byref(mi), C.byref(amd), fmt, boost, C.byref(od), mode, hip.MEM_DEVICE, s) == 0
        st, ref, _ = orc.apply("orc_", oayi, amap, orc.Metadata(omd.maxContentBoost, 1.0, 1.0, 0.0, 0.0, 1.0, omd.maxContentBoost, 1), fmt, boost)
        got = to_host(dout, ref.size)
        if mode == hip.APPLY_EXACT:
            assert np.array_equal(got, ref), (fmt, scale, boost)
        elif fmt == 1:
            assert half_ulp_diff(got.view(np.uint16), ref.view(np.uint16))[0] <= 1
        elif fmt == 4:
            d = np.abs(got.view(np.uint16).astype(np.int32) - ref.view(np.uint16).astype(np.int32))
            assert int((np.minimum(d, 1024 - d) if boost < omd.maxContentBoost else d).max()) <= 1
        else:
            worst, _, ok = diff_1010102(got.view(np.uint32), ref.view(np.uint32), wrap=boost < omd.maxContentBoost)
            assert ok and worst <= 1
        if seed % 3 != 0:   # LUT-mode apply of the same call: bit-exact against the oracle's LUT pipeline
            lout = dev_empty(hip.output_bytes(fmt, aw, ah), 0xCD)
            lod = hip.out_image(lout.data_ptr())
            assert lib.uhdr_hip_apply_gainmap(C.byref(ayi), C.byref(mi), C.byref(amd), fmt, boost, C.byref(lod), hip.APPLY_LUT, hip.MEM_DEVICE, s) == 0
            lst, lref, _ = orc.apply("orc_", oayi, amap, orc.Metadata(omd.maxContentBoost, 1.0, 1.0, 0.0, 0.0, 1.0, omd.maxContentBoost, 1), fmt, boost, lut=True)
            lgot = to_host(lout, lref.size)
            assert lst == 0 and np.array_equal(lgot, lref), ("LUT apply", fmt, scale, boost, aw, ah, int((lgot != lref).sum()))
    # toneMap into a padded destination, convertYuv in place
    dls, dcs = w + int(rng.choice([0, 16, 3])), w // 2 + int(rng.choice([0, 8, 1]))
    d_ty, d_tc = dev_empty(dls * h, 0xEE), dev_empty(dcs * h + dcs, 0xEE)
    tdst = hip.yuv420_image(d_ty.data_ptr(), w, h, -1, dls, dcs, d_tc.data_ptr())
    assert lib.uhdr_hip_tonemap(C.byref(pi), C.byref(tdst), hip.MEM_DEVICE, s) == 0
    oy, oc = np.full(dls * h, 0xEE, np.uint8), np.full(dcs * h + dcs, 0xEE, np.uint8)
    otd = orc.yuv420_image(oy, w, h, -1, dls, dcs, oc)
    assert L.orc_toneMap(C.byref(opi), C.byref(otd)) == 0
    assert np.array_equal(to_host(d_ty, dls * h), oy) and np.array_equal(to_host(d_tc, dcs * h + dcs), oc)
    src, dst = int(rng.randint(0, 3)), int(rng.randint(0, 3))
    assert lib.uhdr_hip_convert_yuv(C.byref(yi), src, dst, hip.MEM_DEVICE, s) == 0
    cyl, cyc = yl.copy(), yc.copy()
    ci = orc.yuv420_image(cyl, w, h, sg, ls, ycs, cyc)
    assert L.orc_convertYuv(C.byref(ci), src, dst) == 0
    assert np.array_equal(to_host(d_yl)[off:off + yl.size], cyl) and np.array_equal(to_host(d_yc)[off:off + yc.size], cyc)
