"""The JPEG files and raw planes the reference's own codec tests run on (tests/golden/: minnie-320x240*.jpg, minnie-*.yu12 / .y,
jpeg_image.jpg -- data files of /root/reference/tests/data, committed as fixtures), through the same calls with the same
expectations, plus what the reference's tests cannot see: the decoded planes and the encoded bytes against libjpeg itself.

  reference test                                         here
  jpegdecoderhelper_test.cpp:96-137 decodeYuvImage,      uhdr_hip_jpeg_decode of the three files: 320 x 240, 4:2:0 / grey, the planes
    decodeYuvIccImage (a PROGRESSIVE file with EXIF,       libjpeg decodes; the ICC profile of the -icc file reads as BT.709, the plain
    XMP and ICC), decodeGreyImage, getCompressedImage-     files carry none (UNSPECIFIED)
    Parameters[Icc]
  jpegencoderhelper_test.cpp:98-123 encodeAligned /      uhdr_hip_jpeg_encode at the test's quality 90 of the 320 x 240, the 318 x 240 and the
    Unaligned / SingleChannelImage                          single-plane image: the bytes libjpeg writes
  jpegr_test.cpp:1815-1900, 2010-2090 EncodeAPI2/3-      encodeJPEGR API-2 / API-3 on the reference's raw 1280 x 720 pair with jpeg_image.jpg as
    AndDecodeTest (jpeg_image.jpg as the SDR JPEG)          the compressed SDR image, strided inputs give the same file, the file decodes
"""
import ctypes as C
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _read(name):
    with open(os.path.join(GOLD, name), "rb") as f:
        return f.read()


FILES = {"yuv": ("minnie-320x240-yuv.jpg", 20193), "icc": ("minnie-320x240-yuv-icc.jpg", 34266), "grey": ("minnie-320x240-y.jpg", 20193)}


def test_fixture_files_are_the_reference_sizes():
    for name, size in FILES.values():
        assert len(_read(name)) == size                       # jpegdecoderhelper_test.cpp:38-40
    assert len(_read("minnie-320x240.yu12")) == 320 * 240 * 3 // 2 and len(_read("minnie-318x240.yu12")) == 318 * 240 * 3 // 2
    assert len(_read("minnie-320x240.y")) == 320 * 240 and len(_read("jpeg_image.jpg")) == 24430


def test_oracle_reads_the_fixture_files_like_libjpeg(orc):
    """the CPU checker (oracle/jpeg_oracle.c) against the image's libjpeg on the reference's files; the progressive one is
    libjpeg's alone (the checker restates the baseline process)"""
    from oracle import jpegr_oracle as J
    for key, (name, _) in FILES.items():
        data = _read(name)
        st, want, w, h, gray = orc.jpeg_decode("lj", data)
        assert st > 0 and (w, h) == (320, 240) and bool(gray) == (key == "grey")        # :117-121 IMAGE_WIDTH / IMAGE_HEIGHT
        if key != "icc":
            st2, got, w2, h2, g2 = orc.jpeg_decode("orc", data)
            assert st2 > 0 and (w2, h2, g2) == (w, h, gray) and np.array_equal(got, want)
        icc = J.app_segment(data, 0xE2, J.ICC_ID)
        if key == "icc":
            assert icc is not None and J.gamut_from_icc(icc) == 0                        # :107, :133 ULTRAHDR_COLORGAMUT_BT709
            assert J.extract_exif(data) is not None                                     # :131 getEXIFSize() > 0
        else:
            assert icc is None                                                          # :99, :122 UNSPECIFIED / getICCSize() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("device", [True, False])
def test_gpu_decodes_the_reference_jpegs(hip, orc, device):
    from tests.test_gpu_jpeg import _gpu_decode
    lib = hip.load()
    for key, (name, _) in FILES.items():
        data = _read(name)
        rc, got, desc = _gpu_decode(lib, hip, data, device)
        st, want, w, h, gray = orc.jpeg_decode("lj", data)
        assert rc == 0 and (desc.width, desc.height) == (320, 240) == (w, h), key
        assert (desc.pixelFormat == hip.PIX_FMT_MONOCHROME) == (key == "grey")
        assert np.array_equal(got, want), (key, int((got != want).sum()))
    data = _read("jpeg_image.jpg")
    rc, got, desc = _gpu_decode(lib, hip, data, device)
    st, want, w, h, gray = orc.jpeg_decode("lj", data)
    assert rc == 0 and (desc.width, desc.height) == (1280, 720) and np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("device", [True, False])
def test_gpu_encodes_the_reference_planes_like_libjpeg(hip, orc, device):
    from tests.test_gpu_jpeg import _gpu_encode
    lib = hip.load()
    q = 90                                                    # jpegencoderhelper_test.cpp:42 JPEG_QUALITY
    for name, w, h in (("minnie-320x240.yu12", 320, 240), ("minnie-318x240.yu12", 318, 240)):
        raw = np.frombuffer(_read(name), np.uint8).copy()
        yb, ub = raw[:w * h], raw[w * h:]
        rc, n, got = _gpu_encode(lib, hip, yb, ub, w, h, q, w, w // 2, device)
        want = orc.jpeg_encode("lj", yb, ub, w, h, q)
        assert rc == 0 and n > 0 and got == want, name        # :104 getCompressedImageSize() > 0 -- and libjpeg's bytes
    yb = np.frombuffer(_read("minnie-320x240.y"), np.uint8).copy()
    rc, n, got = _gpu_encode(lib, hip, yb, None, 320, 240, q, 320, 0, device)
    assert rc == 0 and n > 0 and got == orc.jpeg_encode("lj", yb, None, 320, 240, q)


@pytest.mark.gpu
@pytest.mark.parametrize("device", [True, False])
def test_gpu_api2_api3_with_the_reference_sdr_jpeg(hip, orc, device):
    """jpegr_test.cpp EncodeAPI2AndDecodeTest / EncodeAPI3AndDecodeTest: the raw P010 / YUV420 fixture pair, jpeg_image.jpg as
    the compressed SDR image, HLG; inputs with luma / chroma strides give the same file; it decodes.  Beyond the reference's
    assertions: the file equals the restatement's (oracle/jpegr_oracle.py) and the decoded HDR rendition the oracle's"""
    from oracle import jpegr_oracle as J
    from tests.test_jpegr_container import _Enc, _gpu_decode
    w, h = 1280, 720
    p010 = np.frombuffer(_read("raw_p010_image.p010"), np.uint16).copy()
    yuv = np.frombuffer(_read("raw_yuv420_image.yuv420"), np.uint8).copy()
    sdr = _read("jpeg_image.jpg")
    e = _Enc(hip, device)
    pg, sg = hip.CG_BT2100, hip.CG_BT709
    want2 = J.encode_api2(p010, yuv, w, h, sg, pg, sdr, sg, hip.TF_HLG)
    want3 = J.encode_api3(p010, w, h, pg, sdr, sg, hip.TF_HLG)
    rc, got2 = e.run("api2", e.p010(p010, w, h, pg), e.yuv(yuv, w, h, sg), sdr, sg, hip.TF_HLG)
    assert rc == 0 and got2 == want2
    rc, got3 = e.run("api3", e.p010(p010, w, h, pg), sdr, sg, hip.TF_HLG)
    assert rc == 0 and got3 == want3
    # luma stride + 128, chroma stride + 256 with a chroma plane of its own (the reference's setImageStride / setChromaMode(false))
    ls, cs = w + 128, w + 256
    luma = np.zeros(ls * h, np.uint16)
    luma.reshape(h, ls)[:, :w] = p010[:w * h].reshape(h, w)
    chroma = np.zeros(cs * (h // 2), np.uint16)
    chroma.reshape(h // 2, cs)[:, :w] = p010[w * h:].reshape(h // 2, w)
    if device:
        from tests.gpu_util import to_dev
        dl, dc = to_dev(luma), to_dev(chroma)
        pim = hip.p010_image(dl.data_ptr(), w, h, pg, ls, cs, dc.data_ptr())
    else:
        pim = hip.p010_image(luma.ctypes.data, w, h, pg, ls, cs, chroma.ctypes.data)
    rc, s2 = e.run("api2", pim, e.yuv(yuv, w, h, sg), sdr, sg, hip.TF_HLG)
    assert rc == 0 and s2 == got2
    rc, s3 = e.run("api3", pim, sdr, sg, hip.TF_HLG)
    assert rc == 0 and s3 == got3
    # the files decode (the reference: decodeJPEGR == NO_ERROR), and to what the oracle makes of them
    for blob in (got2, got3):
        rc, dec, dest, md = _gpu_decode(e.lib, hip, blob, hip.OUTPUT_HDR_HLG, 3.4028234663852886e38, hip.APPLY_EXACT, device)
        st, ref, rw, rh, rgamut, rmd = J.decode(blob, hip.OUTPUT_HDR_HLG, 3.4028234663852886e38)
        assert rc == 0 and st == 0 and (dest.width, dest.height, dest.colorGamut) == (w, h, rgamut) == (rw, rh, rgamut)
        assert np.array_equal(dec, np.asarray(ref).view(np.uint8).ravel())


@pytest.mark.gpu
@pytest.mark.parametrize("mono", [False, True])
def test_gpu_effects_on_the_reference_planes_with_the_reference_parameters(hip, orc, mono):
    """editorhelper_test.cpp:98-510 on minnie-320x240.yu12 / .y: crop (10, 99, 20, 199), both mirrors, the three rotations,
    resize to 3/2 and to 2/3 -- the output the oracle gives (itself pinned to the reference's editorhelper.cpp object code),
    with the dimensions the reference asserts"""
    from tests.gpu_util import to_dev, dev_empty, to_host, stream_ptr
    lib, L = hip.load(), orc.load()
    w, h = 320, 240
    buf = np.frombuffer(_read("minnie-320x240.y" if mono else "minnie-320x240.yu12"), np.uint8).copy()
    fmt = orc.FMT_MONOCHROME if mono else orc.FMT_YUV420
    cases = [("crop", (10, 99, 20, 199), (90, 180)), ("mirror", (0,), (w, h)), ("mirror", (1,), (w, h)), ("rotate", (90,), (h, w)),
             ("rotate", (180,), (w, h)), ("rotate", (270,), (h, w)), ("resize", (w * 3 // 2, h * 3 // 2), (480, 360)),
             ("resize", (w * 2 // 3, h * 2 // 3), (213, 160))]
    nbytes = 480 * 360 * 2
    d_buf = to_dev(buf)
    for name, args, dims in cases:
        o_out = np.full(nbytes, 0xCC, np.uint8)
        o_in = orc.Image(buf.ctypes.data, w, h, 1, None, 0, 0, fmt)
        o_img = orc.Image(o_out.ctypes.data, 0, 0, -1, None, 0, 0, -1)
        assert getattr(L, "orc_" + name)(C.byref(o_in), *args, C.byref(o_img)) == 0
        d_out = dev_empty(nbytes, 0xCC)
        g_in = hip.Image(d_buf.data_ptr(), w, h, 1, None, 0, 0, fmt)
        g_img = hip.out_image(d_out.data_ptr())
        assert getattr(lib, "uhdr_hip_" + name)(C.byref(g_in), *args, C.byref(g_img), hip.MEM_DEVICE, stream_ptr()) == 0
        assert (g_img.width, g_img.height) == dims == (o_img.width, o_img.height), name
        assert np.array_equal(to_host(d_out, nbytes), o_out), (name, args)
