"""CPU tests (no GPU): the oracle (oracle/uhdr_oracle.c) is pinned to the reference by

 1. the md5s the REAL reference (ultrahdr.cpp + gainmapmath.cpp compiled verbatim)
    produced on its own 1280x720 test frames (tests/golden/raw_*.p010|yuv420 are those frames);
 2. the checksums the real reference produced on the LCG synthetic frames;
 3. the known-answer values of the reference's tests/gainmapmath_test.cpp, restated;
 4. oracle/_ref -- the reference's gainmapmath.cpp compiled in place -- function by function on random
    inputs, whenever that library exists (it is built only where /root/reference is mounted).
"""
import ctypes as C
import hashlib
import math

import numpy as np
import pytest

FLT_MAX = 3.4028234663852886e38
EPS = 1e-4          # ComparisonEpsilon, gainmapmath_test.cpp:29
YUV_EPS = 1.0 / 510  # YuvConversionEpsilon, :31


def md5(a):
    return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()


GOLDEN_720P = {
    "HLG": dict(tf=1, boost=4.92610836, map="32e38116ea48d76872b525663d33d137", f16="237aa8455a4d5b18136f36e2c59eb8c3",
                pq="87ffff4015d36d3fd664bc918039c00d", hlg="dfd56dc878636c93b7a92b45cf35ee53"),
    "PQ": dict(tf=2, boost=49.2610855, map="83b1c43142efebcedc9d9178b7713aa6", f16="774d53f1befd175102baa1b0597a2082",
               pq="af4217af4760fb92a1cfe7380fe8562a", hlg="4f7d2d8442c48aba6d26f0716ba24611"),
}


def _prefixes(orc):
    return ["orc_"] + (["ref_"] if orc.load_ref() is not None else [])


@pytest.mark.parametrize("tfname", ["HLG", "PQ"])
def test_fixture_md5s_of_the_real_reference(orc, fixture_720p, tfname):
    p010, yuv, w, h = fixture_720p
    g = GOLDEN_720P[tfname]
    for pref in _prefixes(orc):
        yi = orc.yuv420_image(yuv, w, h, orc.CG_BT709)
        pi = orc.p010_image(p010, w, h, orc.CG_BT2100)
        st, m, md = orc.generate(pref, yi, pi, g["tf"], threads=8)
        assert st == 0 and md5(m) == g["map"]
        assert abs(md.maxContentBoost - g["boost"]) < 1e-6 and md.minContentBoost == 1.0
        if tfname == "HLG":   # a saturated HLG pixel encodes as 254, PQ as 255
            assert (int(m.min()), int(m.max())) == (0, 254)
        else:
            assert (int(m.min()), int(m.max())) == (0, 255)
        for fmt, key in ((orc.OUT_HDR_LINEAR, "f16"), (orc.OUT_HDR_PQ, "pq"), (orc.OUT_HDR_HLG, "hlg")):
            st, out, dest = orc.apply(pref, yi, m, md, fmt, FLT_MAX, threads=8)
            assert st == 0 and md5(out) == g[key], (pref, key)
            assert (dest.width, dest.height, dest.colorGamut) == (w, h, orc.CG_BT709)


def test_fixture_tonemap_md5(orc, fixture_720p):
    p010, _, w, h = fixture_720p
    lib = orc.load()
    out = np.full(w * h * 3 // 2, 0xAA, np.uint8)
    src = orc.p010_image(p010, w, h, orc.CG_BT2100)
    dst = orc.yuv420_image(out, w, h, orc.CG_UNSPECIFIED)
    assert lib.orc_toneMap(C.byref(src), C.byref(dst)) == 0
    assert md5(out) == "f8a112ef5d54c8df357fd082b22c5397" and dst.colorGamut == orc.CG_BT2100


LCG_GOLDEN = {
    (640, 480, 1): ("4baa71c620f8c58c", "8b23c9d4ef52f40f", "46e3f5bd53ca8081"),
    (3840, 2160, 1): ("f863201c194c1467", "e87c8f6a91c5e62b", "8228c9d4cdcdf93f"),
    (3840, 2160, 2): ("17d3a7b4d3b562bc", "ba0f86ac7a37a249", "7708aab03d375c7b"),
}


@pytest.mark.parametrize("key", sorted(LCG_GOLDEN))
def test_lcg_checksums_of_the_real_reference(orc, key):
    w, h, tf = key
    lib = orc.load()
    p010, yuv = orc.lcg_frame(w, h, 1234)
    yi, pi = orc.yuv420_image(yuv, w, h, orc.CG_BT709), orc.p010_image(p010, w, h, orc.CG_BT2100)
    st, m, md = orc.generate("orc_", yi, pi, tf, threads=8)
    assert st == 0 and "%016x" % lib.orc_checksum_u8(m.ctypes.data, m.size) == LCG_GOLDEN[key][0]
    for fmt, gi in ((orc.OUT_HDR_HLG, 1), (orc.OUT_HDR_PQ, 2)):
        st, out, _ = orc.apply("orc_", yi, m, md, fmt, FLT_MAX, threads=8)
        assert st == 0 and "%016x" % lib.orc_checksum_u32(out.ctypes.data, out.size // 4) == LCG_GOLDEN[key][gi]


def test_threading_does_not_change_bytes(orc):
    w, h = 320, 240
    p010, yuv = orc.lcg_frame(w, h, 7)
    yi, pi = orc.yuv420_image(yuv, w, h, 0), orc.p010_image(p010, w, h, 2)
    ref = None
    for t in (1, 2, 4, 7):
        st, m, md = orc.generate("orc_", yi, pi, 1, threads=t)
        st2, out, _ = orc.apply("orc_", yi, m, md, orc.OUT_HDR_PQ, FLT_MAX, threads=t)
        cur = (md5(m), md5(out))
        ref = ref or cur
        assert cur == ref


# ---------------------------------------------------------------------------------------------------
# known answers restated from the reference's tests/gainmapmath_test.cpp
# ---------------------------------------------------------------------------------------------------
def _c(orc, r, g, b):
    return orc.Color(r, g, b)


def test_known_answers_transfer_functions(orc):
    L = orc.load()
    f32 = lambda x: float(np.float32(x))
    assert L.orc_srgbInvOetf(0.0) == 0.0 and L.orc_srgbInvOetf(1.0) == 1.0         # :467-473
    assert abs(L.orc_srgbInvOetf(0.02) - 0.00154) < EPS and abs(L.orc_srgbInvOetf(0.04045) - 0.00313) < EPS
    assert abs(L.orc_srgbInvOetf(0.5) - 0.21404) < EPS
    assert L.orc_hlgOetf(0.0) == 0.0 and L.orc_hlgOetf(1.0) == 1.0                 # :752-762
    for x, y in ((0.04167, 0.35357), (0.08333, 0.5), (0.5, 0.87164)):
        assert abs(L.orc_hlgOetf(x) - y) < EPS and abs(L.orc_hlgInvOetf(L.orc_hlgOetf(x)) - x) < EPS
    assert L.orc_hlgInvOetf(0.0) == 0.0
    for x, y in ((0.25, 0.02083), (0.5, 0.08333), (0.75, 0.26496)):             # :764-774
        assert abs(L.orc_hlgInvOetf(x) - y) < EPS
    # HLG white is one ULP above 1.0 in the reference (passes EXPECT_FLOAT_EQ)
    assert np.float32(L.orc_hlgInvOetf(1.0)).view(np.uint32) == 0x3F800001
    assert L.orc_pqOetf(0.0) == 0.0 and L.orc_pqOetf(1.0) == 1.0                   # :784-794
    for x, y in ((0.01, 0.50808), (0.5, 0.92655), (0.99, 0.99895)):
        assert abs(L.orc_pqOetf(x) - y) < EPS and abs(L.orc_pqInvOetf(L.orc_pqOetf(x)) - x) < EPS
    assert L.orc_pqInvOetf(0.0) == 0.0 and L.orc_pqInvOetf(1.0) == 1.0             # :796-806
    for x, y in ((0.01, 2.31017e-7), (0.5, 0.00922), (0.99, 0.90903)):
        assert abs(L.orc_pqInvOetf(x) - y) < EPS


def test_known_answers_luminance_and_yuv(orc):
    L = orc.load()
    lum = {0: (0.2126, 0.7152, 0.0722), 1: (0.20949, 0.72160, 0.06891), 2: (0.2627, 0.6780, 0.0593)}
    for g, (r_, g_, b_) in lum.items():                                          # :408-414,475-481,533-539
        assert L.orc_luminance(g, _c(orc, 0, 0, 0)) == 0.0
        assert abs(L.orc_luminance(g, _c(orc, 1, 1, 1)) - 1.0) < 1e-6
        for col, v in (((1, 0, 0), r_), ((0, 1, 0), g_), ((0, 0, 1), b_)):
            assert np.float32(L.orc_luminance(g, _c(orc, *col))) == np.float32(v)
    prim = {0: ((0.2126, -0.11457, 0.5), (0.7152, -0.38543, -0.45415), (0.0722, 0.5, -0.04585)),
            1: ((0.299, -0.16874, 0.5), (0.587, -0.33126, -0.41869), (0.114, 0.5, -0.08131)),
            2: ((0.2627, -0.13963, 0.5), (0.6780, -0.36037, -0.45979), (0.0593, 0.5, -0.04021))}
    rgbs = ((1, 0, 0), (0, 1, 0), (0, 0, 1))
    for g in (0, 1, 2):                                                          # :416-590
        for rgb, yuvv in zip(rgbs, prim[g]):
            got = L.orc_yuvToRgb(g, _c(orc, *yuvv)).tup()
            assert all(abs(a - b) < EPS for a, b in zip(got, rgb))
            back = L.orc_rgbToYuv(g, _c(orc, *rgb)).tup()
            assert all(abs(a - b) < EPS for a, b in zip(back, yuvv))
        assert L.orc_yuvToRgb(g, _c(orc, 1, 0, 0)).tup() == (1.0, 1.0, 1.0)
    # YUV->YUV matrices map primaries of one encoding onto the other's (:592-692)
    for s in (0, 1, 2):
        for d in (0, 1, 2):
            if s == d:
                continue
            for rgb, want in zip(rgbs, prim[d]):
                got = L.orc_yuvToYuv(s, d, L.orc_rgbToYuv(s, _c(orc, *rgb))).tup()
                assert all(abs(a - b) < YUV_EPS for a, b in zip(got, want)), (s, d)


def test_known_answers_gamut_lookup(orc):                                       # :949-978
    L = orc.load()
    n = C.c_int(0)
    e = _c(orc, 0.25, 0.5, 0.75)
    for s in (-1, 0, 1, 2):
        for h in (-1, 0, 1, 2):
            out = L.orc_gamutConv(s, h, e, C.byref(n)).tup()
            assert n.value == (1 if (s == -1 or h == -1) else 0)
            if s == h and s != -1:
                assert out == e.tup()   # identityConversion


def test_known_answers_encode_gain(orc):                                        # :980-1038 (exact bytes)
    L = orc.load()
    cases = [
        (0.25, 4.0, [(0, 0, 127), (0, 1, 127), (1, 0, 0), (0.5, 0, 0), (1, 1, 127), (1, 4, 255), (1, 5, 255), (4, 1, 0),
                     (4, 0.5, 0), (1, 2, 191), (2, 1, 63)]),
        (0.5, 2.0, [(1, 2, 255), (2, 1, 0), (1, 1.41421, 191), (1.41421, 1, 63)]),
        (0.125, 8.0, [(1, 8, 255), (8, 1, 0), (1, 2.82843, 191), (2.82843, 1, 63)]),
        (1.0, 8.0, [(0, 0, 0), (1, 0, 0), (1, 1, 0), (1, 8, 255), (1, 4, 170), (1, 2, 85)]),
        (0.5, 8.0, [(0, 0, 63), (1, 0, 0), (1, 1, 63), (1, 8, 255), (1, 4, 191), (1, 2, 127), (1, 0.7071, 31), (1, 0.5, 0)]),
    ]
    for mn, mx, rows in cases:
        for sdr, hdr, want in rows:
            assert L.orc_encodeGain3(sdr, hdr, mn, mx) == want, (mn, mx, sdr, hdr)
    # saturation encodes as 254 for HLG's 1000/203 and 255 for PQ's 10000/203
    hl, pq = float(np.float32(1000.0) / np.float32(203.0)), float(np.float32(10000.0) / np.float32(203.0))
    assert L.orc_encodeGain3(1.0, 100.0, 1.0, hl) == 254 and L.orc_encodeGain3(1.0, 100.0, 1.0, pq) == 255


def test_known_answers_apply_gain(orc):                                         # :1040-1113
    L = orc.load()
    white = _c(orc, 1, 1, 1)
    rows = [(0.25, 4.0, [(0.0, 0.25), (0.25, 0.5), (0.5, 1.0), (0.75, 2.0), (1.0, 4.0)]),
            (0.5, 2.0, [(0.0, 0.5), (0.25, 1 / 1.41421), (0.5, 1.0), (0.75, 1.41421), (1.0, 2.0)]),
            (0.125, 8.0, [(0.0, 0.125), (0.25, 1 / 2.82843), (0.5, 1.0), (0.75, 2.82843), (1.0, 8.0)]),
            (1.0, 8.0, [(0.0, 1.0), (1 / 3.0, 2.0), (2 / 3.0, 4.0), (1.0, 8.0)]),
            (0.5, 8.0, [(0.0, 0.5), (0.25, 1.0), (0.5, 2.0), (0.75, 4.0), (1.0, 8.0)])]
    for mn, mx, cases in rows:
        for gain, scale in cases:
            got = L.orc_applyGain3(white, gain, mn, mx).tup()
            assert all(abs(v - scale) < EPS * max(1.0, scale) for v in got), (mn, mx, gain)
        assert L.orc_applyGain3(_c(orc, 0, 0, 0), 0.5, mn, mx).tup() == (0.0, 0.0, 0.0)
    e = _c(orc, 0.0, 0.5, 1.0)
    for col in (white, e, _c(orc, 1, 0, 0)):    # applyGain(e,1,md) == applyGain(e,1,md,displayBoost=max) (EXPECT_RGB_EQ)
        a, b = L.orc_applyGain3(col, 1.0, 0.25, 4.0).tup(), L.orc_applyGain4(col, 1.0, 0.25, 4.0, 4.0).tup()
        assert all(abs(x - y) <= 4 * np.spacing(np.float32(max(abs(x), 1e-30))) for x, y in zip(a, b))


def _img4x4(orc):
    ypix = np.array([0x00, 0x10, 0x20, 0x30, 0x01, 0x11, 0x21, 0x31, 0x02, 0x12, 0x22, 0x32, 0x03, 0x13, 0x23, 0x33,
                     0xA0, 0xA1, 0xA2, 0xA3, 0xB0, 0xB1, 0xB2, 0xB3], np.uint8)
    ppix = np.array([0x00, 0x10, 0x20, 0x30, 0x01, 0x11, 0x21, 0x31, 0x02, 0x12, 0x22, 0x32, 0x03, 0x13, 0x23, 0x33,
                     0xA0, 0xB0, 0xA1, 0xB1, 0xA2, 0xB2, 0xA3, 0xB3], np.uint16) << 6
    return ypix, ppix


def test_known_answers_pixel_fetch_and_sampling(orc):                           # :106-229,1115-1187
    L = orc.load()
    ypix, ppix = _img4x4(orc)
    yi = orc.Image(ypix.ctypes.data, 4, 4, 0, ypix.ctypes.data + 16, 4, 2, 1)
    pi = orc.Image(ppix.ctypes.data, 4, 4, 0, ppix.ctypes.data + 32, 4, 4, 0)
    k255, k876, k896 = np.float32(1) / np.float32(255), np.float32(1) / np.float32(876), np.float32(1) / np.float32(896)
    for y in range(4):
        for x in range(4):
            yy = (x << 4) | y
            u, v = 0xA0 + (x // 2) + 2 * (y // 2), 0xB0 + (x // 2) + 2 * (y // 2)
            got = L.orc_getYuv420Pixel(C.byref(yi), x, y).tup()
            want = (np.float32(yy) * k255, np.float32(u - 128) * k255, np.float32(v - 128) * k255)
            assert got == tuple(float(t) for t in want)
            got = L.orc_getP010Pixel(C.byref(pi), x, y).tup()
            want = (np.float32(yy - 64) * k876, np.float32(u - 64) * k896 - np.float32(0.5), np.float32(v - 64) * k896 - np.float32(0.5))
            assert got == tuple(float(t) for t in want)
    for img, fn, get in ((yi, L.orc_sampleYuv420, L.orc_getYuv420Pixel), (pi, L.orc_sampleP010, L.orc_getP010Pixel)):
        for y in range(2):
            for x in range(2):
                px = [get(C.byref(img), 2 * x + dx, 2 * y + dy).tup() for dy in range(2) for dx in range(2)]
                s = fn(C.byref(img), 2, x, y).tup()
                for ch in range(3):     # within min/max of the sampled pixels (:1137-1187)
                    assert min(p[ch] for p in px) - 1e-6 <= s[ch] <= max(p[ch] for p in px) + 1e-6


def test_known_answers_sample_map_table_equals_float(orc):                      # :1189-1227 (EXPECT_EQ)
    L = orc.load()
    rng = np.random.RandomState(0)
    m = rng.randint(0, 256, (4, 4)).astype(np.uint8)
    mi = orc.map_image(m, 4, 4)
    for scale in (2, 4):
        for y in range(4 * scale):
            for x in range(4 * scale):
                a = L.orc_sampleMapIdw(C.byref(mi), scale, x, y)
                b = L.orc_sampleMapFloat(C.byref(mi), float(scale), x, y)
                xb, yb = x // scale, y // scale
                nb = [m[yy, xx] / 255.0 for yy in (yb, min(yb + 1, 3)) for xx in (xb, min(xb + 1, 3))]
                assert min(nb) - 1e-6 <= a <= max(nb) + 1e-6
                if scale == 2:
                    assert a == b, (x, y)


def test_known_answers_packing(orc):                                            # :1229-1262 (exact)
    L = orc.load()
    assert L.orc_colorToRgba1010102(_c(orc, 0, 0, 0)) == 0x3 << 30
    assert L.orc_colorToRgba1010102(_c(orc, 1, 1, 1)) == 0xFFFFFFFF
    assert L.orc_colorToRgba1010102(_c(orc, 1, 0, 0)) == (0x3 << 30 | 0x3ff)
    assert L.orc_colorToRgba1010102(_c(orc, 0, 1, 0)) == (0x3 << 30 | 0x3ff << 10)
    assert L.orc_colorToRgba1010102(_c(orc, 0, 0, 1)) == (0x3 << 30 | 0x3ff << 20)
    f = lambda v: int(np.float32(v) * np.float32(1023))
    assert L.orc_colorToRgba1010102(_c(orc, 0.1, 0.2, 0.3)) == (0x3 << 30 | f(0.1) | f(0.2) << 10 | f(0.3) << 20)
    assert L.orc_colorToRgbaF16(_c(orc, 0, 0, 0)) == 0x3C00 << 48
    assert L.orc_colorToRgbaF16(_c(orc, 1, 1, 1)) == 0x3C003C003C003C00
    assert L.orc_colorToRgbaF16(_c(orc, 0.1, 0.2, 0.3)) == 0x3C0034CD32662E66
    for v, want in ((0.1, 0x2E66), (0.0, 0x0), (1.0, 0x3C00), (-1.0, 0xBC00), (FLT_MAX, 0x7FFF), (-FLT_MAX, 0xFFFF),
                    (1.1754943508222875e-38, 0x0)):
        assert L.orc_floatToHalf(v) == want


def test_known_answers_transform_yuv420_block(orc):                             # :694-750
    L = orc.load()
    for s, d in ((0, 1), (0, 2), (1, 0), (1, 2), (2, 0), (2, 1)):
        ypix, _ = _img4x4(orc)
        src = ypix.copy()
        img = orc.Image(ypix.ctypes.data, 4, 4, s, ypix.ctypes.data + 16, 4, 2, 1)
        simg = orc.Image(src.ctypes.data, 4, 4, s, src.ctypes.data + 16, 4, 2, 1)
        L.orc_transformYuv420(C.byref(img), 1, 1, s, d)
        for y in range(4):
            for x in range(4):
                if x >= 2 and y >= 2:
                    continue    # all other pixels stay unchanged
                assert L.orc_getYuv420Pixel(C.byref(simg), x, y).tup() == L.orc_getYuv420Pixel(C.byref(img), x, y).tup()
        px = [L.orc_yuvToYuv(s, d, L.orc_getYuv420Pixel(C.byref(simg), 2 + dx, 2 + dy)).tup() for dy in range(2) for dx in range(2)]
        for i, (dx, dy) in enumerate(((0, 0), (1, 0), (0, 1), (1, 1))):
            got = L.orc_getYuv420Pixel(C.byref(img), 2 + dx, 2 + dy).tup()
            assert abs(got[0] - px[i][0]) < YUV_EPS
            assert abs(got[1] - sum(p[1] for p in px) / 4) < YUV_EPS
            assert abs(got[2] - sum(p[2] for p in px) / 4) < YUV_EPS


# ---------------------------------------------------------------------------------------------------
# function-by-function against the reference's own compiled gainmapmath.cpp (when available)
# ---------------------------------------------------------------------------------------------------
def test_restatement_equals_reference_object_code(orc):
    R = orc.load_ref()
    if R is None:
        pytest.skip("oracle/_ref not built (needs /root/reference); the md5/checksum pins above still hold")
    L = orc.load()
    rng = np.random.RandomState(123)
    xs = np.concatenate([rng.uniform(0, 1, 20000), rng.uniform(0, 0.1, 5000), [0, 1, 0.04045, 0.5, 1 / 12, 1e-4, 0.0001]]).astype(np.float32)
    for name in ("srgbInvOetf", "hlgOetf", "hlgInvOetf", "pqOetf", "pqInvOetf"):
        fo, fr = getattr(L, "orc_" + name), getattr(R, "ref_" + name)
        for x in xs:
            a, b = fo(float(x)), fr(float(x))
            assert a == b or (math.isnan(a) and math.isnan(b)), (name, x)
    cols = rng.uniform(-0.2, 1.2, (3000, 3)).astype(np.float32)
    n = C.c_int()
    for c in cols:
        col = orc.Color(*[float(t) for t in c])
        for g in (0, 1, 2):
            assert L.orc_luminance(g, col) == R.ref_luminance(g, col)
            assert L.orc_yuvToRgb(g, col).tup() == R.ref_yuvToRgb(g, col).tup()
            assert L.orc_rgbToYuv(g, col).tup() == R.ref_rgbToYuv(g, col).tup()
            for h in (0, 1, 2):
                assert L.orc_gamutConv(g, h, col, C.byref(n)).tup() == R.ref_gamutConv(g, h, col, C.byref(n)).tup()
                assert L.orc_yuvToYuv(g, h, col).tup() == R.ref_yuvToYuv(g, h, col).tup()
        assert L.orc_colorToRgba1010102(col) == R.ref_colorToRgba1010102(col) or c.min() < 0 or c.max() > 1.0009
        assert L.orc_colorToRgbaF16(col) == R.ref_colorToRgbaF16(col)
        g_ = float(abs(c[0]) % 1.0)
        for mn, mx, db in ((1.0, 4.926108, 4.926108), (0.5, 8.0, 3.0), (1.0, 49.26108, 49.26108)):
            assert L.orc_applyGain3(col, g_, mn, mx).tup() == R.ref_applyGain3(col, g_, mn, mx).tup()
            assert L.orc_applyGain4(col, g_, mn, mx, db).tup() == R.ref_applyGain4(col, g_, mn, mx, db).tup()
        ys, yh = float(abs(c[1]) * 203), float(abs(c[2]) * 1000)
        for mn, mx in ((1.0, 4.926108), (0.25, 4.0), (1.0, 49.26108)):
            assert L.orc_encodeGain3(ys, yh, mn, mx) == R.ref_encodeGain3(ys, yh, mn, mx)
    for v in np.concatenate([rng.uniform(-70000, 70000, 3000), rng.uniform(-1e-4, 1e-4, 3000), [0, 65504, 65520, 1e-8]]).astype(np.float32):
        assert L.orc_floatToHalf(float(v)) == R.ref_floatToHalf(float(v))
    for scale in (1, 2, 3, 4, 5, 8):
        for incR, incB in ((1, 1), (0, 1), (1, 0), (0, 0)):
            a = (C.c_float * (scale * scale * 4))()
            b = (C.c_float * (scale * scale * 4))()
            L.orc_fillShepardsIDW(a, scale, incR, incB)
            R.ref_fillShepardsIDW(b, scale, incR, incB)
            assert list(a) == list(b)


def test_whole_image_functions_equal_reference_object_code(orc):
    R = orc.load_ref()
    if R is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    L = orc.load()
    w, h = 96, 64
    p010, yuv = orc.lcg_frame(w, h, 55)
    for sg in (0, 1, 2):
        for hg in (0, 1, 2):
            for tf in (0, 1, 2):
                for is601 in (False, True):
                    yi, pi = orc.yuv420_image(yuv, w, h, sg), orc.p010_image(p010, w, h, hg)
                    a = orc.generate("orc_", yi, pi, tf, is601, threads=1)
                    b = orc.generate("ref_", yi, pi, tf, is601, threads=1)
                    assert a[0] == b[0] == 0 and np.array_equal(a[1], b[1]), (sg, hg, tf, is601)
    yi = orc.yuv420_image(yuv, w, h, 0)
    for scale, mw, mh in ((4, 24, 16), (2, 48, 32), (8, 12, 8), (1, 96, 64)):
        m = np.random.RandomState(scale).randint(0, 256, (mh, mw)).astype(np.uint8)
        md = orc.Metadata(6.0, 0.5, 1.0, 0.0, 0.0, 0.5, 6.0, 1)
        for fmt in (1, 2, 3, 4):
            for boost in (FLT_MAX, 3.0):
                a = orc.apply("orc_", yi, m, md, fmt, boost, threads=2)
                b = orc.apply("ref_", yi, m, md, fmt, boost, threads=1)
                assert a[0] == b[0] == 0 and np.array_equal(a[1], b[1]), (scale, fmt, boost)
    for s, d in ((0, 1), (0, 2), (1, 0), (1, 2), (2, 0), (2, 1)):
        a, b = yuv.copy(), yuv.copy()
        ia, ib = orc.yuv420_image(a, w, h, s), orc.yuv420_image(b, w, h, s)
        assert L.orc_convertYuv(C.byref(ia), s, d) == 0 and R.ref_convertYuv(C.byref(ib), s, d) == 0
        assert np.array_equal(a, b)


def test_error_codes_follow_the_reference_order(orc):                           # ultrahdr.cpp:189-202,364-406
    L = orc.load()
    w, h = 16, 8
    p010, yuv = orc.lcg_frame(w, h, 1)
    yi, pi = orc.yuv420_image(yuv, w, h, 0), orc.p010_image(p010, w, h, 2)
    md, out = orc.Metadata(), np.zeros(64, np.uint8)
    call = lambda y, p, tf: L.orc_generateGainMap(y, p, tf, C.byref(md), out.ctypes.data, 0, 1)
    assert call(None, C.byref(pi), 1) == -10001
    p2 = orc.p010_image(p010, w + 2, h, 2)
    assert call(C.byref(yi), C.byref(p2), 1) == -10006
    y2 = orc.yuv420_image(yuv, w, h, -1)
    assert call(C.byref(y2), C.byref(pi), 1) == -10003
    assert call(C.byref(yi), C.byref(pi), 3) == -10005
    m = np.zeros((2, 4), np.uint8)
    good = orc.Metadata(4.0, 1.0, 1.0, 0.0, 0.0, 1.0, 4.0, 1)
    assert orc.apply("orc_", yi, m, good, 2, 4.0)[0] == 0
    bad = orc.Metadata(4.0, 1.0, 1.0, 0.0, 0.0, 1.0, 4.0, 0)
    assert orc.apply("orc_", yi, m, bad, 2, 4.0)[0] == -10010
    bad = orc.Metadata(4.0, 1.0, 2.2, 0.0, 0.0, 1.0, 4.0, 1)
    assert orc.apply("orc_", yi, m, bad, 2, 4.0)[0] == -10010
    bad = orc.Metadata(4.0, 1.0, 1.0, 0.0, 0.0, 1.0, 5.0, 1)
    assert orc.apply("orc_", yi, m, bad, 2, 4.0)[0] == -10010
    assert orc.apply("orc_", yi, np.zeros((2, 3), np.uint8), good, 2, 4.0)[0] == -20008
    assert orc.apply("orc_", yi, np.zeros((4, 4), np.uint8), good, 2, 4.0)[0] == -20008


# ---------------------------------------------------------------------------------------------------
# editorhelper effects against the reference's own editorhelper.cpp object code
# ---------------------------------------------------------------------------------------------------
def _fx_image(orc, rng, w, h, mono, ls=None, cs=None, sep_chroma=False):
    ls = ls or w
    cs = cs or ls // 2
    luma = rng.randint(0, 256, ls * h).astype(np.uint8)
    if mono:
        return (luma,), orc.Image(luma.ctypes.data, w, h, 0, None, ls, 0, orc.FMT_MONOCHROME)
    if sep_chroma:
        chroma = rng.randint(0, 256, cs * h).astype(np.uint8)
        return (luma, chroma), orc.Image(luma.ctypes.data, w, h, 1, chroma.ctypes.data, ls, cs, orc.FMT_YUV420)
    buf = rng.randint(0, 256, ls * h + cs * h).astype(np.uint8)
    return (buf,), orc.Image(buf.ctypes.data, w, h, 1, None, ls, cs if cs != ls // 2 else 0, orc.FMT_YUV420)


def _fx_cases():
    cases = []
    for mono in (False, True):
        cases += [("crop", mono, (10, 53, 4, 31)), ("crop", mono, (0, 63, 0, 39)), ("crop", mono, (2, 2, 6, 7)),
                  ("mirror", mono, (0,)), ("mirror", mono, (1,)), ("rotate", mono, (90,)), ("rotate", mono, (180,)),
                  ("rotate", mono, (270,)), ("resize", mono, (128, 80)), ("resize", mono, (32, 20)), ("resize", mono, (50, 34))]
    return cases


def _fx_run(orc, lib, prefix, name, img, args, out_bytes):
    out = np.full(out_bytes, 0xCC, np.uint8)
    oimg = orc.Image(out.ctypes.data, 0, 0, -1, None, 0, 0, -1)
    rc = getattr(lib, prefix + name)(C.byref(img), *args, C.byref(oimg))
    return rc, out, (oimg.width, oimg.height, oimg.colorGamut, oimg.luma_stride, oimg.chroma_stride, oimg.pixelFormat,
                     (oimg.chroma_data or 0) - out.ctypes.data if oimg.chroma_data else None)


@pytest.mark.parametrize("layout", ["tight", "strided", "separate_chroma"])
def test_effects_restatement_equals_reference_object_code(orc, layout):
    R = orc.load_ref()
    if R is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    L = orc.load()
    rng = np.random.RandomState(5)
    w, h = 64, 40
    for name, mono, args in _fx_cases():
        ls = w + 6 if layout != "tight" else None
        cs = (w + 6) // 2 + 3 if layout == "separate_chroma" else None
        keep, img = _fx_image(orc, rng, w, h, mono, ls, cs, sep_chroma=(layout == "separate_chroma"))
        nbytes = 4 * (max(w, 128) + 8) * (max(h, 80) + 8)
        a = _fx_run(orc, L, "orc_", name, img, args, nbytes)
        b = _fx_run(orc, R, "ref_", name, img, args, nbytes)
        assert a[0] == b[0] == 0, (name, args, a[0], b[0])
        assert a[2] == b[2], (name, mono, args, a[2], b[2])
        assert np.array_equal(a[1], b[1]), (name, mono, args)


def test_effects_error_codes(orc):                                              # editorhelper.cpp:29-39,175-185
    L = orc.load()
    rng = np.random.RandomState(1)
    keep, img = _fx_image(orc, rng, 32, 16, False)
    out = np.zeros(4096, np.uint8)
    o = orc.Image(out.ctypes.data, 0, 0, -1, None, 0, 0, -1)
    assert L.orc_crop(None, 0, 1, 0, 1, C.byref(o)) == -10001
    assert L.orc_crop(C.byref(img), -1, 5, 0, 5, C.byref(o)) == -10011
    assert L.orc_crop(C.byref(img), 0, 32, 0, 5, C.byref(o)) == -10011
    assert L.orc_crop(C.byref(img), 0, 5, 0, 16, C.byref(o)) == -10011
    assert L.orc_rotate(C.byref(img), 45, C.byref(o)) == -10011
    p010 = orc.Image(img.data, 32, 16, 0, None, 0, 0, orc.FMT_P010)
    assert L.orc_mirror(C.byref(p010), 0, C.byref(o)) == -30000
    assert L.orc_resize(C.byref(p010), 8, 8, C.byref(o)) == -30000


# ---------------------------------------------------------------------------------------------------
# LUT variants: restated reference tests + the reference's own object code
# ---------------------------------------------------------------------------------------------------
LUTS = (("srgbInvOetf", 0, 1024), ("hlgInvOetf", 1, 4096), ("pqInvOetf", 2, 4096), ("hlgOetf", 4, 65536), ("pqOetf", 5, 65536))


def test_known_answers_lut_accessors_at_the_knots(orc):                         # gainmapmath_test.cpp:808-841
    L = orc.load()
    for name, which, n in LUTS:
        knots = (np.arange(n, dtype=np.float32) / np.float32(n - 1)).astype(np.float32)
        direct = orc.eval_transfer({0: 0, 1: 1, 2: 2, 4: 4, 5: 5}[which], knots)
        via_lut = orc.eval_transfer(40 + which, knots)
        assert np.array_equal(direct, via_lut), name                           # EXPECT_FLOAT_EQ holds with equality
        assert np.array_equal(orc.lut_table(which), direct), name
        assert getattr(L, "orc_" + name + "LUT")(float(knots[n // 3])) == direct[n // 3]


def test_known_answers_apply_gain_lut(orc):                                     # gainmapmath_test.cpp:843-939
    L = orc.load()
    colors = [(0, 0, 0), (1, 1, 1), (1, 0, 0), (0, 1, 0), (0, 0, 1)]
    values = (np.arange(1024, dtype=np.float32) / np.float32(1023)).astype(np.float32)
    for boost in range(1, 11):
        for mn in (1.0 / boost, 1.0, 1.0 / float(np.float32(boost) ** np.float32(1.0 / 3.0))):
            mn, mx = float(np.float32(mn)), float(boost)
            plain = orc.gain_lut("orc_", mn, mx)
            with_boost = orc.gain_lut("orc_", mn, mx, mx)
            assert np.array_equal(plain, with_boost)                           # EXPECT_RGB_EQ
            tbl = plain.ctypes.data_as(C.POINTER(C.c_float))
            for v in values[::37]:
                for c in colors:
                    col = orc.Color(*[float(t) for t in c])
                    a = L.orc_applyGain3(col, float(v), mn, mx).tup()
                    b = L.orc_applyGainLUT(col, float(v), tbl).tup()
                    assert max(abs(p - q) for p, q in zip(a, b)) <= EPS        # EXPECT_RGB_NEAR


def test_lut_restatement_equals_reference_object_code(orc):
    R = orc.load_ref()
    if R is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    L = orc.load()
    rng = np.random.RandomState(7)
    xs = np.concatenate([rng.uniform(0, 1, 30000), rng.uniform(0.999, 1.3, 2000), rng.uniform(0, 2e-3, 3000),
                         [0, 1, 0.5, 2.0, 70000.0, 4.3e9, 1e19, 3e38, -0.0001, -0.3, -5.0, float("nan"), float("inf")]]).astype(np.float32)
    for name, which, n in LUTS:
        fo, fr = getattr(L, "orc_" + name + "LUT"), getattr(R, "ref_" + name + "LUT")
        half = (np.arange(n, dtype=np.float64) + 0.5) / (n - 1)                  # index rounding boundaries
        pts = np.concatenate([xs, half.astype(np.float32)[:: max(1, n // 4096)],
                              np.nextafter(half.astype(np.float32), np.float32(0))[:: max(1, n // 4096)]])
        for x in pts:
            assert fo(float(x)) == fr(float(x)), (name, float(x))
    for mn, mx, db in ((1.0, 4.926108, 4.926108), (0.5, 8.0, 3.0), (1.0, 49.26108, 49.26108), (0.25, 4.0, 1.0), (1.0, 1.0, 1.0),
                       (1.0, 4.926108, FLT_MAX), (2.0, 16.0, 0.0)):
        assert np.array_equal(orc.gain_lut("orc_", mn, mx), orc.gain_lut("ref_", mn, mx)), (mn, mx)
        assert np.array_equal(orc.gain_lut("orc_", mn, mx, db), orc.gain_lut("ref_", mn, mx, db)), (mn, mx, db)
        t = orc.gain_lut("orc_", mn, mx, db)
        tbl = t.ctypes.data_as(C.POINTER(C.c_float))
        for g in rng.uniform(0, 1, 500).astype(np.float32):
            assert L.orc_gainLutFactor(tbl, float(g)) == R.ref_gainLutFactor(mn, mx, db, float(g))


def test_lut_pipelines_equal_reference_object_code(orc):
    """generateGainMap / applyGainMap with the USE_*_LUT branches (ultrahdr.cpp:230,238,319,433,446,470,481) taken,
    restatement vs the reference's own LUT functions driven in the same order"""
    R = orc.load_ref()
    if R is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    w, h = 96, 64
    p010, yuv = orc.lcg_frame(w, h, 77)
    for sg, hg in ((0, 2), (1, 1), (2, 0)):
        for tf in (0, 1, 2):
            yi, pi = orc.yuv420_image(yuv, w, h, sg), orc.p010_image(p010, w, h, hg)
            a = orc.generate("orc_", yi, pi, tf, False, threads=2, lut=True)
            b = orc.generate("ref_", yi, pi, tf, False, threads=1, lut=True)
            assert a[0] == b[0] == 0 and np.array_equal(a[1], b[1]), (sg, hg, tf)
            plain = orc.generate("orc_", yi, pi, tf, False, threads=2)
            assert np.abs(a[1].astype(int) - plain[1].astype(int)).max() <= (3 if tf else 1)   # a LUT is an approximation
    yi = orc.yuv420_image(yuv, w, h, 0)
    for scale, mw, mh in ((4, 24, 16), (2, 48, 32), (1, 96, 64)):
        m = np.random.RandomState(scale).randint(0, 256, (mh, mw)).astype(np.uint8)
        for md in (orc.Metadata(6.0, 0.5, 1.0, 0.0, 0.0, 0.5, 6.0, 1), orc.Metadata(4.926108, 1.0, 1.0, 0.0, 0.0, 1.0, 4.926108, 1)):
            for fmt in (1, 2, 3, 4):
                for boost in (FLT_MAX, 3.0):
                    a = orc.apply("orc_", yi, m, md, fmt, boost, threads=2, lut=True)
                    b = orc.apply("ref_", yi, m, md, fmt, boost, threads=1, lut=True)
                    assert a[0] == b[0] == 0 and np.array_equal(a[1], b[1]), (scale, fmt, boost)


# addEffects (editorhelper.cpp:362-446): the chain as ultrahdr.cpp applies it to the SDR image and to the gain map
FX_CHAINS = [
    [],
    [(3, 96, 72, 0, 0)],
    [(3, 96, 72, 0, 0), (1, 0, 0, 0, 0), (2, 90, 0, 0, 0), (0, 10, 49, 20, 83)],      # the shape of editorhelper_test.cpp:553-600
    [(0, 4, 99, 2, 61), (2, 180, 0, 0, 0), (1, 1, 0, 0, 0), (3, 64, 40, 0, 0), (2, 270, 0, 0, 0)],
    [(2, 90, 0, 0, 0), (2, 90, 0, 0, 0), (2, 90, 0, 0, 0), (2, 90, 0, 0, 0)],
]


def _fx_array(orc, chain):
    arr = (orc.Effect * max(len(chain), 1))()
    for i, e in enumerate(chain):
        arr[i] = orc.Effect(*e)
    return arr


@pytest.mark.parametrize("mono", [False, True])
def test_add_effects_restatement_equals_reference_object_code(orc, mono):
    R = orc.load_ref()
    if R is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    L = orc.load()
    rng = np.random.RandomState(5)
    w, h = 128, 96
    keep, img = _fx_image(orc, rng, w, h, mono)
    for chain in FX_CHAINS:
        arr = _fx_array(orc, chain)
        outs = []
        for lib, pre in ((L, "orc_"), (R, "ref_")):
            buf = np.full(w * h * 2 + 64, 0xEE, np.uint8)
            o = orc.Image(buf.ctypes.data, 0, 0, -1, None, 0, 0, -1)
            rc = getattr(lib, pre + "add_effects")(C.byref(img), arr, len(chain), C.byref(o))
            outs.append((rc, buf, (o.width, o.height, o.colorGamut, o.pixelFormat, o.luma_stride, o.chroma_stride),
                         None if (mono or not chain) else o.chroma_data - o.data))
        assert outs[0][0] == outs[1][0] == 0, chain
        assert outs[0][2:] == outs[1][2:], (chain, outs[0][2:], outs[1][2:])
        assert np.array_equal(outs[0][1], outs[1][1]), chain
    # defined deviation: the reference ignores a failing effect's status (its own test passes clockwise_degree = 900,
    # editorhelper_test.cpp:571, and then copies uninitialised fields); the restatement reports it
    buf = np.zeros(w * h * 2, np.uint8)
    o = orc.Image(buf.ctypes.data, 0, 0, -1, None, 0, 0, -1)
    assert L.orc_add_effects(C.byref(img), _fx_array(orc, [(2, 900, 0, 0, 0)]), 1, C.byref(o)) == -10011
    assert L.orc_add_effects(None, arr, 0, C.byref(o)) == -10001
