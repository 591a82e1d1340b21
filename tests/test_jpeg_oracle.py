"""CPU tests (no GPU): the JPEG checker oracle/jpeg_oracle.c -- a restatement of what the reference's
JpegEncoderHelper::compressImage (lib/src/jpegencoderhelper.cpp:39-283) makes libjpeg write -- is pinned to

 1. the image's own libjpeg (IJG 9d) driven with the helper's call sequence (oracle/jpeg_libjpeg_harness.c): whole files,
    byte for byte, for every size class (MCU-aligned, ragged width, ragged height, tiny), both strides regimes (zero
    padding vs. reading the caller's padding columns), qualities 1..100, YUV 4:2:0 and single plane, with and without ICC;
 2. Pillow's bundled libjpeg-turbo (the family the reference pins, 3.0.1) for MCU-aligned sizes: Pillow cannot be fed raw
    4:2:0 planes, but chroma replicated 2x2 downsamples back to itself exactly in libjpeg's h2v2 box filter;
 3. a decode with Pillow: what comes back is the input within JPEG loss.

The reference's tests (tests/jpegencoderhelper_test.cpp) only check that encoding succeeds and yields a non-empty
buffer; those properties are restated at the end.
"""
import io
import os
import subprocess
import sys

import numpy as np
import pytest


def _content(kind, w, h, rng):
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == "smooth":
        y = np.clip((np.sin(xx / 7.0) + np.cos(yy / 5.0)) * 50 + 128 + rng.randint(-6, 7, (h, w)), 0, 255)
        u = (xx[::2, ::2] * 2 + 30) % 256
        v = (yy[::2, ::2] * 5 + 90) % 256
    elif kind == "noise":
        y, u, v = rng.randint(0, 256, (h, w)), rng.randint(0, 256, (h // 2, w // 2)), rng.randint(0, 256, (h // 2, w // 2))
    elif kind == "extreme":   # long zero runs, ZRL codes, maximum-magnitude coefficients, 0xFF bytes in the stream
        y = np.where((xx // 3 + yy // 5) % 2 == 0, 0, 255)
        u = np.where(xx[::2, ::2] % 7 == 0, 255, 0)
        v = np.full((h // 2, w // 2), 255)
    else:                     # flat: DC-only blocks
        y, u, v = np.full((h, w), 77), np.full((h // 2, w // 2), 128), np.full((h // 2, w // 2), 200)
    return y.astype(np.uint8), u.astype(np.uint8), v.astype(np.uint8)


def _planes(y, u, v, ls, cs, rng):
    """luma buffer with stride ls, chroma buffer (U then V at cs*h/2) with stride cs; padding columns hold random bytes"""
    h, w = y.shape
    yb = rng.randint(0, 256, (h, ls)).astype(np.uint8)
    yb[:, :w] = y
    ub = rng.randint(0, 256, (h, cs)).astype(np.uint8)      # h/2 rows of U followed by h/2 rows of V, +slack
    ub[:h // 2, :w // 2] = u
    ub[h // 2:h // 2 * 2, :w // 2] = v
    return np.ascontiguousarray(yb), np.ascontiguousarray(ub)


SIZES = [(64, 48), (16, 16), (48, 32), (40, 24), (24, 40), (34, 18), (2, 2), (18, 2), (2, 18), (130, 66), (256, 144)]


@pytest.mark.parametrize("kind", ["smooth", "noise", "extreme", "flat"])
def test_oracle_equals_libjpeg_behind_the_reference_call_sequence(orc, kind):
    if orc.load_libjpeg() is None:
        pytest.skip("no libjpeg in this image")
    rng = np.random.RandomState(len(kind))
    for w, h in SIZES:
        y, u, v = _content(kind, w, h, rng)
        aw, acw = (w + 15) // 16 * 16, (w // 2 + 7) // 8 * 8
        for ls, cs in ((w, w // 2), (aw, acw), (aw + 16, acw + 8), (w + 2, w // 2 + 1)):
            yb, ub = _planes(y, u, v, ls, cs, rng)
            for q in (90, 85, 50, 20, 1, 100) if (w, h) in ((64, 48), (40, 24)) else (85,):
                a = orc.jpeg_encode("orc", yb, ub, w, h, q, ls, cs)
                b = orc.jpeg_encode("lj", yb, ub, w, h, q, ls, cs)
                assert a == b, ("yuv420", kind, w, h, ls, cs, q, len(a), len(b))
                a = orc.jpeg_encode("orc", yb, None, w, h, q, ls)
                b = orc.jpeg_encode("lj", yb, None, w, h, q, ls)
                assert a == b, ("plane", kind, w, h, ls, q, len(a), len(b))
    # ICC payload as APP2 right after the JFIF header (jpegencoderhelper.cpp:98-100)
    y, u, v = _content("smooth", 64, 48, rng)
    yb, ub = _planes(y, u, v, 64, 32, rng)
    icc = bytes(range(200)) * 3
    a, b = orc.jpeg_encode("orc", yb, ub, 64, 48, 90, icc=icc), orc.jpeg_encode("lj", yb, ub, 64, 48, 90, icc=icc)
    assert a == b and a[20:24] == b"\xff\xe2\x02\x5a"


_PIL_SCRIPT = r"""
import io, sys, numpy as np
from PIL import Image
d = np.load(sys.argv[1])
out = {}
for key in d.files:
    if not key.startswith("y_"): continue
    tag = key[2:]
    w, h, q = [int(t) for t in tag.split("_")]
    y, u, v = d["y_" + tag], d["u_" + tag], d["v_" + tag]
    ycc = np.stack([y, np.repeat(np.repeat(u, 2, 0), 2, 1), np.repeat(np.repeat(v, 2, 0), 2, 1)], -1)
    b = io.BytesIO(); Image.fromarray(ycc, mode="YCbCr").save(b, "JPEG", quality=q, subsampling=2, optimize=False)
    out["c_" + tag] = np.frombuffer(b.getvalue(), np.uint8)
    b = io.BytesIO(); Image.fromarray(y, mode="L").save(b, "JPEG", quality=q, optimize=False)
    out["g_" + tag] = np.frombuffer(b.getvalue(), np.uint8)
np.savez(sys.argv[2], **out)
"""


def test_oracle_equals_pillow_libjpeg_turbo_on_mcu_aligned_sizes(orc, tmp_path):
    """Pillow runs in its own process: it bundles its own libjpeg, which must not share a process with the harness's"""
    try:
        import PIL  # noqa: F401
    except ImportError:
        pytest.skip("Pillow not installed")
    rng = np.random.RandomState(3)
    cases, arrays = [], {}
    for kind, (w, h, q) in zip(("smooth", "noise", "extreme", "smooth", "noise"), ((64, 48, 90), (128, 96, 85), (256, 144, 50), (48, 32, 100), (32, 16, 3))):
        y, u, v = _content(kind, w, h, rng)
        tag = "%d_%d_%d" % (w, h, q)
        arrays.update({"y_" + tag: y, "u_" + tag: u, "v_" + tag: v})
        cases.append((tag, y, u, v, w, h, q))
    np.savez(tmp_path / "in.npz", **arrays)
    subprocess.check_call([sys.executable, "-c", _PIL_SCRIPT, str(tmp_path / "in.npz"), str(tmp_path / "out.npz")])
    res = np.load(tmp_path / "out.npz")
    for tag, y, u, v, w, h, q in cases:
        uv = np.ascontiguousarray(np.concatenate([u.reshape(-1), v.reshape(-1)]))
        assert orc.jpeg_encode("orc", np.ascontiguousarray(y), uv, w, h, q) == res["c_" + tag].tobytes(), tag
        assert orc.jpeg_encode("orc", np.ascontiguousarray(y), None, w, h, q) == res["g_" + tag].tobytes(), tag


def test_quant_tables_and_header_layout(orc):
    lib = orc.load()
    q = np.zeros(64, np.uint16)
    lib.orc_jpeg_quant_table(50, 0, q.ctypes.data)      # quality 50 = the Annex K table itself
    assert q[:8].tolist() == [16, 11, 10, 16, 24, 40, 51, 61] and q[63] == 99
    lib.orc_jpeg_quant_table(100, 1, q.ctypes.data)
    assert (q == 1).all()
    lib.orc_jpeg_quant_table(1, 0, q.ctypes.data)       # force_baseline clamps to 255
    assert q.max() == 255
    hdr = np.zeros(2048, np.uint8)
    n = lib.orc_jpeg_header(3840, 2160, 0, 95, None, 0, hdr.ctypes.data, hdr.size)
    b = hdr[:n].tobytes()
    assert b[:4] == b"\xff\xd8\xff\xe0" and b[6:11] == b"JFIF\0" and b.count(b"\xff\xdb") == 2 and b.count(b"\xff\xc4") == 4
    assert b[-14:-12] == b"\xff\xda" and n == 2 + 18 + 2 * 69 + 19 + (33 + 183) * 2 + 14
    n1 = lib.orc_jpeg_header(960, 540, 1, 85, None, 0, hdr.ctypes.data, hdr.size)
    assert n1 == 2 + 18 + 69 + 13 + 33 + 183 + 10


def test_encoder_properties_the_reference_tests_check(orc):
    """tests/jpegencoderhelper_test.cpp: valid input encodes to a non-empty buffer; here additionally: it decodes back"""
    rng = np.random.RandomState(11)
    w, h = 320, 240
    y, u, v = _content("smooth", w, h, rng)
    uv = np.ascontiguousarray(np.concatenate([u.reshape(-1), v.reshape(-1)]))
    data = orc.jpeg_encode("orc", np.ascontiguousarray(y), uv, w, h, 90)
    assert len(data) > 0 and data[:2] == b"\xff\xd8" and data[-2:] == b"\xff\xd9"
    script = ("import io,sys,numpy as np\nfrom PIL import Image\nim=Image.open(io.BytesIO(open(sys.argv[1],'rb').read()));im.draft('YCbCr',None)\n"
              "a=np.asarray(im.convert('YCbCr'))[:,:,0].astype(int);np.save(sys.argv[2],a)")
    import os
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "a.jpg"), "wb").write(data)
        try:
            subprocess.check_call([sys.executable, "-c", script, os.path.join(d, "a.jpg"), os.path.join(d, "y.npy")])
        except subprocess.CalledProcessError:
            pytest.skip("Pillow not usable")
        back = np.load(os.path.join(d, "y.npy"))
    assert back.shape == (h, w) and np.abs(back - y.astype(int)).mean() < 3.0


# ---------------------------------------------------------------------------------------------------------------------
# decoding: JpegDecoderHelper::decompressImage(..., DECODE_TO_YCBCR) (lib/src/jpegdecoderhelper.cpp:188-516)
# ---------------------------------------------------------------------------------------------------------------------
_PIL_ENCODE = r"""
import io, sys, numpy as np
from PIL import Image
d = np.load(sys.argv[1]); out = {}
y, u, v = d["y"], d["u"], d["v"]
ycc = np.stack([y, np.repeat(np.repeat(u, 2, 0), 2, 1), np.repeat(np.repeat(v, 2, 0), 2, 1)], -1)
im = Image.fromarray(ycc, mode="YCbCr")
def enc(img, **kw):
    b = io.BytesIO(); img.save(b, "JPEG", **kw); return np.frombuffer(b.getvalue(), np.uint8)
out["opt"] = enc(im, quality=80, subsampling=2, optimize=True)                       # non-default Huffman tables
out["rst"] = enc(im, quality=90, subsampling=2, restart_marker_blocks=5)             # restart intervals
out["rst_rows"] = enc(im, quality=60, subsampling=2, restart_marker_rows=1)
out["gray_opt"] = enc(Image.fromarray(y, mode="L"), quality=85, optimize=True)
out["prog"] = enc(im, quality=80, subsampling=2, progressive=True)                   # outside the restatement
out["s444"] = enc(im, quality=80, subsampling=0)                                     # the reference rejects it
np.savez(sys.argv[2], **out)
"""


def jpeg_corpus(orc, tmp_path=None):
    """(name, jpeg bytes) pairs: the oracle encoder's own output over sizes / qualities / content, and (when Pillow is
    usable) libjpeg-turbo files with optimised tables and restart markers"""
    rng = np.random.RandomState(21)
    corpus = []
    for kind in ("smooth", "noise", "extreme", "flat"):
        for (w, h), q in zip(SIZES, (90, 85, 50, 20, 100, 75, 95, 60, 1, 85, 92)):
            y, u, v = _content(kind, w, h, rng)
            uv = np.ascontiguousarray(np.concatenate([u.reshape(-1), v.reshape(-1)]))
            corpus.append(("%s_%dx%d_q%d" % (kind, w, h, q), orc.jpeg_encode("orc", np.ascontiguousarray(y), uv, w, h, q)))
            corpus.append(("%s_plane_%dx%d_q%d" % (kind, w, h, q), orc.jpeg_encode("orc", np.ascontiguousarray(y), None, w, h, q)))
    extra = {}
    if tmp_path is not None:
        y, u, v = _content("smooth", 208, 112, rng)
        np.savez(tmp_path / "pin.npz", y=y, u=u, v=v)
        try:
            subprocess.check_call([sys.executable, "-c", _PIL_ENCODE, str(tmp_path / "pin.npz"), str(tmp_path / "pout.npz")],
                                  stderr=subprocess.DEVNULL)
            res = np.load(tmp_path / "pout.npz")
            extra = {k: res[k].tobytes() for k in res.files}
        except (subprocess.CalledProcessError, OSError):
            extra = {}
    return corpus, extra


def test_decoder_restatement_equals_libjpeg(orc, tmp_path):
    if orc.load_libjpeg() is None:
        pytest.skip("no libjpeg in this image")
    corpus, extra = jpeg_corpus(orc, tmp_path)
    for name, data in corpus + [(k, extra[k]) for k in ("opt", "rst", "rst_rows", "gray_opt") if k in extra]:
        a = orc.jpeg_decode("orc", data)
        b = orc.jpeg_decode("lj", data)
        assert a[0] == b[0] > 0 and a[2:] == b[2:], (name, a[0], b[0])
        assert np.array_equal(a[1], b[1]), (name, int((a[1] != b[1]).sum()))
    if "prog" in extra:   # libjpeg reads progressive files; the baseline restatement says "unsupported" (the product decodes them: test_jpeg_progressive.py)
        assert orc.jpeg_decode("lj", extra["prog"])[0] > 0 and orc.jpeg_decode("orc", extra["prog"])[0] == -2
    if "s444" in extra:   # jpegdecoderhelper.cpp:256-262: only 4:2:0
        assert orc.jpeg_decode("lj", extra["s444"])[0] == -2 and orc.jpeg_decode("orc", extra["s444"])[0] == -2
    assert orc.jpeg_decode("orc", b"\xff\xd8\xff\xd9")[0] == -1 and orc.jpeg_decode("orc", b"notajpeg")[0] == -1


def test_decode_of_encode_is_close_to_the_input(orc):
    rng = np.random.RandomState(2)
    w, h = 128, 96
    y, u, v = _content("smooth", w, h, rng)
    uv = np.ascontiguousarray(np.concatenate([u.reshape(-1), v.reshape(-1)]))
    st, planes, dw, dh, gray = orc.jpeg_decode("orc", orc.jpeg_encode("orc", np.ascontiguousarray(y), uv, w, h, 95))
    assert st == w * h * 3 // 2 and (dw, dh, gray) == (w, h, 0)
    assert np.abs(planes[:w * h].astype(int) - y.reshape(-1)).mean() < 2.0
    assert np.abs(planes[w * h:w * h * 5 // 4].astype(int) - u.reshape(-1)).mean() < 2.0


def test_rgba_restatement_equals_libjpeg_turbo(orc, tmp_path):
    """orc_ycc420_to_rgba (libjpeg-turbo's DECODE_TO_RGBA path: fancy h2v2 upsampling + fixed-point colour conversion, restated from
    the published algorithm) against libjpeg-turbo itself, which Pillow bundles: every 4:2:0 file of the corpus plus small and narrow
    sizes, and the primary image of the reference's own sample_jpegr.jpeg"""
    try:
        import io
        from PIL import Image, features
    except ImportError:
        pytest.skip("no Pillow in this image")
    if not features.check_feature("libjpeg_turbo"):
        pytest.skip("Pillow without libjpeg-turbo")
    corpus, extra = jpeg_corpus(orc, tmp_path)
    rng = np.random.RandomState(4)
    for (w, h) in ((2, 18), (4, 4), (4, 18), (6, 6), (8, 8), (16, 2), (10, 14), (66, 34), (640, 480)):
        y = rng.randint(0, 256, w * h * 3 // 2).astype(np.uint8)
        corpus.append(("rand_%dx%d" % (w, h), orc.jpeg_encode("orc", y[:w * h], y[w * h:], w, h, 30 + 6 * (w % 11))))
    sample = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sample_jpegr.jpeg"), "rb").read()
    corpus.append(("sample_jpegr primary", sample[:42326]))
    checked = 0
    for name, data in corpus + [(k, extra[k]) for k in ("opt", "rst", "rst_rows") if k in extra]:
        st, planes, w, h, gray = orc.jpeg_decode("orc", data)
        if gray or st <= 0 or (w & 1) or (h & 1):
            continue
        want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
        got = orc.ycc420_to_rgba(planes, w, h)
        assert np.array_equal(got[..., :3], want), (name, int((got[..., :3] != want).sum()))
        assert np.all(got[..., 3] == 255)
        checked += 1
    assert checked >= 40
