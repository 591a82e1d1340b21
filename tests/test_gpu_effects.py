"""-m gpu: editorhelper effects (crop / mirror / rotate / resize) through the C-ABI against the
oracle, which is itself checked against the reference's own editorhelper.cpp object code in tests/test_oracle_pins.py.
Pure byte movement: bit-exact, including the reference's layout quirks."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [("crop", (10, 53, 4, 31)), ("crop", (0, 63, 0, 39)), ("crop", (2, 2, 6, 7)), ("crop", (1, 62, 3, 38)),
         ("mirror", (0,)), ("mirror", (1,)), ("rotate", (90,)), ("rotate", (180,)), ("rotate", (270,)),
         ("resize", (128, 80)), ("resize", (32, 20)), ("resize", (50, 34))]


def _make(rng, w, h, mono, layout):
    ls = w + 6 if layout != "tight" else w
    cs = (w + 6) // 2 + 3 if layout == "separate_chroma" else ls // 2
    luma = rng.randint(0, 256, ls * h).astype(np.uint8)
    chroma = None if mono else rng.randint(0, 256, cs * h).astype(np.uint8)
    return luma, chroma, ls, cs


@pytest.mark.parametrize("layout", ["tight", "strided", "separate_chroma"])
@pytest.mark.parametrize("mono", [False, True])
@pytest.mark.parametrize("mem", ["device", "host"])
def test_effects_match_the_oracle(hip, orc, layout, mono, mem):
    from tests.gpu_util import to_dev, dev_empty, to_host, stream_ptr
    lib, L = hip.load(), orc.load()
    rng = np.random.RandomState(11)
    w, h = 64, 40
    nbytes = 4 * 136 * 88
    for name, args in CASES:
        luma, chroma, ls, cs = _make(rng, w, h, mono, layout)
        if layout == "separate_chroma" or mono:
            buf, cptr_off = luma, None
        else:  # chroma right after luma in one allocation, chroma_data == NULL
            buf, cptr_off = np.concatenate([luma, chroma]), None
        fmt = orc.FMT_MONOCHROME if mono else orc.FMT_YUV420
        # oracle
        o_out = np.full(nbytes, 0xCC, np.uint8)
        o_in = orc.Image(buf.ctypes.data, w, h, 1, chroma.ctypes.data if (layout == "separate_chroma" and not mono) else None,
                         ls if layout != "tight" else 0, cs if layout == "separate_chroma" else 0, fmt)
        o_img = orc.Image(o_out.ctypes.data, 0, 0, -1, None, 0, 0, -1)
        assert getattr(L, "orc_" + name)(C.byref(o_in), *args, C.byref(o_img)) == 0
        # product
        if mem == "device":
            d_buf = to_dev(buf)
            d_chr = to_dev(chroma) if (layout == "separate_chroma" and not mono) else None
            d_out = dev_empty(nbytes, 0xCC)
            g_in = hip.Image(d_buf.data_ptr(), w, h, 1, d_chr.data_ptr() if d_chr is not None else None, o_in.luma_stride,
                             o_in.chroma_stride, fmt)
            g_img = hip.out_image(d_out.data_ptr())
            rc = getattr(lib, "uhdr_hip_" + name)(C.byref(g_in), *args, C.byref(g_img), hip.MEM_DEVICE, stream_ptr())
            got = to_host(d_out, nbytes)
            base = d_out.data_ptr()
        else:
            got = np.full(nbytes, 0xCC, np.uint8)
            g_in = hip.Image(buf.ctypes.data, w, h, 1, o_in.chroma_data, o_in.luma_stride, o_in.chroma_stride, fmt)
            g_img = hip.out_image(got.ctypes.data)
            rc = getattr(lib, "uhdr_hip_" + name)(C.byref(g_in), *args, C.byref(g_img), hip.MEM_HOST, None)
            base = got.ctypes.data
        assert rc == 0, (name, args, rc)
        assert (g_img.width, g_img.height, g_img.colorGamut, g_img.luma_stride, g_img.pixelFormat) == \
               (o_img.width, o_img.height, o_img.colorGamut, o_img.luma_stride, o_img.pixelFormat), (name, args)
        if not mono:
            assert g_img.chroma_stride == o_img.chroma_stride
            assert g_img.chroma_data - base == o_img.chroma_data - o_out.ctypes.data
        assert np.array_equal(got, o_out), (name, args, mono, layout, int((got != o_out).sum()))


def test_effects_error_codes(hip):
    from tests.gpu_util import dev_empty, stream_ptr
    lib = hip.load()
    d_in, d_out = dev_empty(32 * 16 * 3 // 2, 1), dev_empty(4096, 0)
    img = hip.Image(d_in.data_ptr(), 32, 16, 0, None, 0, 0, hip.PIX_FMT_YUV420)
    out = hip.out_image(d_out.data_ptr())
    s = stream_ptr()
    assert lib.uhdr_hip_crop(None, 0, 1, 0, 1, C.byref(out), hip.MEM_DEVICE, s) == hip.ERROR_BAD_PTR
    assert lib.uhdr_hip_crop(C.byref(img), -1, 5, 0, 5, C.byref(out), hip.MEM_DEVICE, s) == hip.ERROR_INVALID_CROPPING_PARAMETERS
    assert lib.uhdr_hip_crop(C.byref(img), 0, 32, 0, 5, C.byref(out), hip.MEM_DEVICE, s) == hip.ERROR_INVALID_CROPPING_PARAMETERS
    assert lib.uhdr_hip_crop(C.byref(img), 0, 5, 0, 16, C.byref(out), hip.MEM_DEVICE, s) == hip.ERROR_INVALID_CROPPING_PARAMETERS
    assert lib.uhdr_hip_rotate(C.byref(img), 45, C.byref(out), hip.MEM_DEVICE, s) == hip.ERROR_INVALID_CROPPING_PARAMETERS
    p010 = hip.Image(d_in.data_ptr(), 32, 16, 0, None, 0, 0, hip.PIX_FMT_P010)
    assert lib.uhdr_hip_mirror(C.byref(p010), 0, C.byref(out), hip.MEM_DEVICE, s) == hip.ERROR_UNSUPPORTED_FEATURE
    assert lib.uhdr_hip_resize(C.byref(p010), 8, 8, C.byref(out), hip.MEM_DEVICE, s) == hip.ERROR_UNSUPPORTED_FEATURE


def test_effects_full_size_roundtrips(hip, orc):
    """4K properties that need no oracle: rotate 90 x4 == identity, mirror twice == identity, rotate 180 == both mirrors"""
    from tests.gpu_util import to_dev, dev_empty, to_host, stream_ptr
    lib = hip.load()
    w, h = 3840, 2160
    _, yuv = orc.lcg_frame(w, h, 77)
    n = w * h * 3 // 2
    a, b = to_dev(yuv), dev_empty(n, 0)
    s = stream_ptr()
    cur = hip.Image(a.data_ptr(), w, h, 1, None, 0, 0, hip.PIX_FMT_YUV420)
    bufs = [b, a]
    for k in range(4):
        out = hip.out_image(bufs[k % 2].data_ptr())
        assert lib.uhdr_hip_rotate(C.byref(cur), 90, C.byref(out), hip.MEM_DEVICE, s) == 0
        cur = hip.Image(out.data, out.width, out.height, out.colorGamut, None, 0, 0, out.pixelFormat)
    assert (cur.width, cur.height) == (w, h)
    assert np.array_equal(to_host(a, n), yuv)
    m1, m2, r180 = dev_empty(n, 0), dev_empty(n, 0), dev_empty(n, 0)
    src = hip.Image(a.data_ptr(), w, h, 1, None, 0, 0, hip.PIX_FMT_YUV420)
    o1 = hip.out_image(m1.data_ptr())
    assert lib.uhdr_hip_mirror(C.byref(src), 0, C.byref(o1), hip.MEM_DEVICE, s) == 0
    i1 = hip.Image(m1.data_ptr(), w, h, 1, None, 0, 0, hip.PIX_FMT_YUV420)
    o2 = hip.out_image(m2.data_ptr())
    assert lib.uhdr_hip_mirror(C.byref(i1), 1, C.byref(o2), hip.MEM_DEVICE, s) == 0
    o3 = hip.out_image(r180.data_ptr())
    assert lib.uhdr_hip_rotate(C.byref(src), 180, C.byref(o3), hip.MEM_DEVICE, s) == 0
    assert np.array_equal(to_host(m2, n), to_host(r180, n))


@pytest.mark.parametrize("mono", [False, True])
@pytest.mark.parametrize("device", [True, False])
def test_add_effects_chains(hip, orc, mono, device):
    """addEffects (editorhelper.cpp:362-446): chains stay on the device; bytes and descriptor equal the oracle's restatement
    (pinned to the reference's object code by tests/test_oracle_pins.py)"""
    from tests.gpu_util import dev_empty, stream_ptr, to_dev, to_host
    from tests.test_oracle_pins import FX_CHAINS, _fx_image
    lib, L = hip.load(), orc.load()
    rng = np.random.RandomState(5 + int(mono))
    w, h = 128, 96
    for ls in (None, 144):
        keep, img = _fx_image(orc, rng, w, h, mono, ls=ls)
        src = keep[0]
        for chain in FX_CHAINS:
            padded_mirror_first = ls is not None and chain and (chain[0][0] == 1 or (chain[0][0] == 2 and chain[0][1] == 180))
            oarr = (orc.Effect * max(len(chain), 1))(*[orc.Effect(*e) for e in chain])
            garr = (hip.Effect * max(len(chain), 1))(*[hip.Effect(*e) for e in chain])
            obuf = np.full(160 * 160 * 2, 0xEE, np.uint8)
            oo = orc.Image(obuf.ctypes.data, 0, 0, -1, None, 0, 0, -1)
            assert L.orc_add_effects(C.byref(img), oarr, len(chain), C.byref(oo)) == 0
            if device:
                dsrc, dout = to_dev(src), dev_empty(obuf.size, 0xEE)
                gi = hip.Image(dsrc.data_ptr(), w, h, img.colorGamut, None, img.luma_stride, img.chroma_stride, img.pixelFormat)
                go = hip.Image(dout.data_ptr(), 0, 0, -1, None, 0, 0, -1)
                rc = lib.uhdr_hip_add_effects(C.byref(gi), garr, len(chain), C.byref(go), hip.MEM_DEVICE, stream_ptr())
                got = to_host(dout, obuf.size)
                base = dout.data_ptr()
            else:
                gbuf = np.full(obuf.size, 0xEE, np.uint8)
                gi = hip.Image(src.ctypes.data, w, h, img.colorGamut, None, img.luma_stride, img.chroma_stride, img.pixelFormat)
                go = hip.Image(gbuf.ctypes.data, 0, 0, -1, None, 0, 0, -1)
                rc = lib.uhdr_hip_add_effects(C.byref(gi), garr, len(chain), C.byref(go), hip.MEM_HOST, None)
                got, base = gbuf, gbuf.ctypes.data
            if padded_mirror_first:
                assert rc == hip.ERROR_UNSUPPORTED_FEATURE
                continue
            assert rc == 0, (chain, rc)
            assert (go.width, go.height, go.colorGamut, go.pixelFormat, go.luma_stride, go.chroma_stride) == \
                   (oo.width, oo.height, oo.colorGamut, oo.pixelFormat, oo.luma_stride, oo.chroma_stride), chain
            if chain and not mono:
                assert go.chroma_data - base == oo.chroma_data - oo.data
            assert np.array_equal(got, obuf), (chain, ls, int((got != obuf).sum()))
    bad = (hip.Effect * 1)(hip.Effect(2, 900, 0, 0, 0))
    keep, img = _fx_image(orc, rng, w, h, mono)
    gbuf = np.zeros(w * h * 2, np.uint8)
    gi = hip.Image(keep[0].ctypes.data, w, h, 0, None, 0, 0, img.pixelFormat)
    go = hip.Image(gbuf.ctypes.data, 0, 0, -1, None, 0, 0, -1)
    assert lib.uhdr_hip_add_effects(C.byref(gi), bad, 1, C.byref(go), hip.MEM_HOST, None) == hip.ERROR_INVALID_CROPPING_PARAMETERS
    assert lib.uhdr_hip_add_effects(None, bad, 1, C.byref(go), hip.MEM_HOST, None) == hip.ERROR_BAD_PTR
