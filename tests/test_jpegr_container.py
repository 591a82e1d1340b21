"""decodeJPEGR (lib/src/jpegr.cpp:655-822): container scan + XMP metadata + ICC gamut + two JPEG decodes + applyGainMap.

CPU part: the Python restatement (oracle/jpegr_oracle.py) on the reference's own sample file (tests/data/sample_jpegr.jpeg,
committed as a fixture) -- properties the file itself fixes.  GPU part: uhdr_hip_jpegr_decode against that restatement, on the
sample and on JPEG/R files assembled here from the device encoder's output.  The reference's decodeJPEGR / encodeJPEGR API-0..3
are not buildable here (ultrahdr.cpp) and its tests keep no decoded bytes; what IS pinned against the reference's own object code:
the container writers and parsers and whole API-4 files (tests/test_ref_container.py), and every stage underneath (JPEG coding,
generateGainMap, applyGainMap) on its own."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLT_MAX = 3.4028234663852886e38
SAMPLE = os.path.join(ROOT, "tests", "golden", "sample_jpegr.jpeg")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _xmp_segment(attrs):
    body = ('<x:xmpmeta\n  xmlns:x="adobe:ns:meta/"\n  x:xmptk="Adobe XMP Core 5.1.2">\n  <rdf:RDF\n    xmlns:rdf="http://www.w3.org/1999/02/22-rdf-syntax-ns#">\n'
            '    <rdf:Description\n      xmlns:hdrgm="http://ns.adobe.com/hdr-gain-map/1.0/"' +
            "".join('\n      hdrgm:%s="%s"' % kv for kv in attrs) + "/>\n  </rdf:RDF>\n</x:xmpmeta>\n").encode()
    payload = b"http://ns.adobe.com/xap/1.0/\0" + body
    return b"\xff\xe1" + struct.pack(">H", len(payload) + 2) + payload


def assemble_jpegr(primary, gainmap, attrs):
    """primary JPEG + gain map JPEG with an XMP APP1 right after its SOI (the layout of jpegr.cpp:951-1130, minus MPF / ICC)"""
    return primary + gainmap[:2] + _xmp_segment(attrs) + gainmap[2:]


GOOD_ATTRS = (("Version", "1.0"), ("GainMapMin", "0"), ("GainMapMax", "2.3"), ("Gamma", "1"), ("OffsetSDR", "0"), ("OffsetHDR", "0"),
              ("HDRCapacityMin", "0"), ("HDRCapacityMax", "2.3"), ("BaseRenditionIsHDR", "False"))


def test_sample_file_container_metadata_and_gamut(orc):
    from oracle import jpegr_oracle as J
    data = open(SAMPLE, "rb").read()
    assert J.find_images(data) == [(0, 42326), (42326, 3727)]          # Item:Length="3727" in the primary image's own XMP
    st, out, w, h, gamut, md = J.decode(data, orc.OUT_HDR_HLG, FLT_MAX)
    assert st == 0 and (w, h) == (1280, 720) and gamut == orc.CG_BT709  # sRGB colorants in the embedded ICC profile
    assert md["version"] == "1.0" and abs(float(md["max"]) - 10.0) < 1e-4 and md["min"] == 1.0 and md["capmax"] == md["max"]
    assert out.size == w * h * 4 and (out.view(np.uint32) >> 30 == 3).all()
    assert J.decode(data[:42326], orc.OUT_HDR_HLG, FLT_MAX)[0] == -20003 and J.decode(b"junk", orc.OUT_HDR_HLG, FLT_MAX)[0] == -20006
    assert J.decode(data, orc.OUT_HDR_HLG, 0.5)[0] == -10008
    # defaults and refusals of getMetadataFromXMP (jpegrutils.cpp:499-545)
    md = J.metadata_from_xmp(_xmp_segment((("Version", "1.0"), ("GainMapMax", "1"), ("HDRCapacityMax", "1")))[4:])
    assert md["min"] == 1.0 and md["gamma"] == 1.0 and md["off_sdr"] == np.float32(1 / 64) and md["capmin"] == 1.0 and md["max"] == 2.0
    assert J.metadata_from_xmp(_xmp_segment((("Version", "1.0"), ("GainMapMax", "1")))[4:]) is None
    assert J.metadata_from_xmp(_xmp_segment(GOOD_ATTRS[:-1] + (("BaseRenditionIsHDR", "True"),))[4:]) is None
    assert J.metadata_from_xmp(_xmp_segment((("Version", "1.0"), ("GainMapMax", "x"), ("HDRCapacityMax", "1")))[4:]) is None


def _gpu_decode(lib, hip, data, fmt, boost, mode, device):
    from tests.gpu_util import dev_empty, stream_ptr, to_host
    buf = np.frombuffer(data, np.uint8)
    dest, md = hip.Image(), hip.Metadata()
    rc = lib.uhdr_hip_jpegr_decode(C.c_void_p(buf.ctypes.data), buf.size, fmt, boost, None, 0, C.byref(dest), C.byref(md), mode, hip.MEM_HOST, None)
    if rc != hip.ERROR_INSUFFICIENT_RESOURCE:
        return rc, None, dest, md
    need = hip.output_bytes(fmt, dest.width, dest.height)
    if device:
        d = dev_empty(need, 0xCD)
        rc = lib.uhdr_hip_jpegr_decode(C.c_void_p(buf.ctypes.data), buf.size, fmt, boost, C.c_void_p(d.data_ptr()), need, C.byref(dest), C.byref(md), mode,
                                       hip.MEM_DEVICE, stream_ptr())
        return rc, (to_host(d, need).copy() if rc == 0 else None), dest, md
    out = np.full(need, 0xCD, np.uint8)
    rc = lib.uhdr_hip_jpegr_decode(C.c_void_p(buf.ctypes.data), buf.size, fmt, boost, C.c_void_p(out.ctypes.data), need, C.byref(dest), C.byref(md), mode, hip.MEM_HOST, None)
    return rc, (out if rc == 0 else None), dest, md


@pytest.mark.gpu
@pytest.mark.parametrize("device", [True, False])
def test_gpu_decodes_the_reference_sample_file(hip, orc, device):
    from oracle import jpegr_oracle as J
    from tests.gpu_util import diff_1010102
    lib = hip.load()
    data = open(SAMPLE, "rb").read()
    for fmt in (hip.OUTPUT_HDR_HLG, hip.OUTPUT_HDR_PQ, hip.OUTPUT_HDR_LINEAR, hip.OUTPUT_HDR_LINEAR_RGB_10BIT):
        for boost in (FLT_MAX, 4.0):
            st, want, w, h, gamut, omd = J.decode(data, fmt, boost)
            rc, got, dest, md = _gpu_decode(lib, hip, data, fmt, boost, hip.APPLY_EXACT, device)
            assert rc == st == 0 and (dest.width, dest.height, dest.colorGamut) == (w, h, gamut)
            assert md.version == b"1.0" and md.maxContentBoost == omd["max"] and md.hdrCapacityMax == omd["capmax"] and md.minContentBoost == 1.0
            assert np.array_equal(got, want), (fmt, boost, int((got != want).sum()))
        rc, fast, _, _ = _gpu_decode(lib, hip, data, fmt, FLT_MAX, hip.APPLY_FAST, device)
        assert rc == 0
        if fmt in (hip.OUTPUT_HDR_HLG, hip.OUTPUT_HDR_PQ):
            want = J.decode(data, fmt, FLT_MAX)[1]
            worst, frac, alpha_ok = diff_1010102(fast.view(np.uint32), want.view(np.uint32))
            assert alpha_ok and worst <= 1


@pytest.mark.gpu
def test_gpu_decodes_files_assembled_from_the_device_encoder(hip, orc):
    """generate a gain map on the device, compress frame and map there, glue them with an XMP packet, decode the result"""
    from oracle import jpegr_oracle as J
    from tests.gpu_util import gpu_generate, to_dev
    from tests.test_gpu_jpeg import _gpu_encode
    from tests.test_gpu_parity import smooth_frame
    lib = hip.load()
    w, h = 512, 256
    p010, yuv = smooth_frame(w, h, 3)
    dp, dy = to_dev(p010), to_dev(yuv)
    st, gmap, gmd, _ = gpu_generate(lib, hip.yuv420_image(dy.data_ptr(), w, h, hip.CG_BT709), hip.p010_image(dp.data_ptr(), w, h, hip.CG_BT2100), hip.TF_HLG)
    assert st == 0
    rc, _, pj = _gpu_encode(lib, hip, yuv[:w * h], yuv[w * h:], w, h, 95, w, w // 2, True)
    rc2, _, gj = _gpu_encode(lib, hip, np.ascontiguousarray(gmap.reshape(-1)), None, w // 4, h // 4, 85, w // 4, 0, True)
    assert rc == 0 and rc2 == 0
    l2 = "%.6g" % np.log2(np.float64(gmd.maxContentBoost))
    attrs = (("Version", "1.0"), ("GainMapMin", "0"), ("GainMapMax", l2), ("Gamma", "1"), ("OffsetSDR", "0"), ("OffsetHDR", "0"),
             ("HDRCapacityMin", "0"), ("HDRCapacityMax", l2), ("BaseRenditionIsHDR", "False"))
    data = assemble_jpegr(pj, gj, attrs)
    for fmt in (hip.OUTPUT_HDR_HLG, hip.OUTPUT_HDR_LINEAR):
        st, want, ow, oh, gamut, omd = J.decode(data, fmt, FLT_MAX)
        rc, got, dest, md = _gpu_decode(lib, hip, data, fmt, FLT_MAX, hip.APPLY_EXACT, True)
        assert rc == st == 0 and (dest.width, dest.height) == (w, h) and dest.colorGamut == hip.CG_UNSPECIFIED   # no ICC segment
        assert np.array_equal(got, want)
    # statuses in the reference's order
    assert _gpu_decode(lib, hip, pj, hip.OUTPUT_HDR_HLG, FLT_MAX, 0, True)[0] == -20003                     # GAIN_MAP_IMAGE_NOT_FOUND
    assert _gpu_decode(lib, hip, b"no jpeg in here at all", hip.OUTPUT_HDR_HLG, FLT_MAX, 0, True)[0] == -20006  # NO_IMAGES_FOUND
    assert _gpu_decode(lib, hip, data, hip.OUTPUT_HDR_HLG, 0.5, 0, True)[0] == -10008                       # INVALID_DISPLAY_BOOST
    assert _gpu_decode(lib, hip, data, 9, FLT_MAX, 0, True)[0] == -10009                                    # INVALID_OUTPUT_FORMAT
    assert _gpu_decode(lib, hip, pj + gj, hip.OUTPUT_HDR_HLG, FLT_MAX, 0, True)[0] == -20005                # METADATA_ERROR (no XMP)
    bad = assemble_jpegr(pj, gj, attrs[:3] + (("Gamma", "2.2"),) + attrs[4:])
    assert _gpu_decode(lib, hip, bad, hip.OUTPUT_HDR_HLG, FLT_MAX, 0, True)[0] == hip.ERROR_BAD_METADATA       # applyGainMap's own check
    # the SDR rendition works without a readable gain map or XMP packet (jpegr.cpp:728,754: neither is touched)
    # ... unless the caller asks for the metadata (:754-760)
    assert _gpu_decode(lib, hip, pj + gj, hip.OUTPUT_SDR, FLT_MAX, 0, True)[0] == -20005
    plain = np.frombuffer(pj + gj, np.uint8)
    sdest = hip.Image()
    st, want_sdr, sw, sh, _, _ = J.decode(pj + gj, orc.OUT_SDR, FLT_MAX)
    sdr = np.zeros(sw * sh * 4, np.uint8)
    rc = lib.uhdr_hip_jpegr_decode(C.c_void_p(plain.ctypes.data), plain.size, hip.OUTPUT_SDR, FLT_MAX, C.c_void_p(sdr.ctypes.data), sdr.size, C.byref(sdest), None,
                                   hip.APPLY_FAST, hip.MEM_HOST, None)
    assert rc == st == 0 and (sdest.width, sdest.height) == (sw, sh) and np.array_equal(sdr, want_sdr)
    assert _gpu_decode(lib, hip, gj + pj, hip.OUTPUT_HDR_HLG, FLT_MAX, 0, True)[0] == -20002                # primary is not 4:2:0: DECODE_ERROR


# ---------------------------------------------------------------------------------------------------------------------
# assembly side (encodeJPEGR API-1)
# ---------------------------------------------------------------------------------------------------------------------
def _sample_streams():
    """the two JPEG streams as they were before appendGainMap wrapped them: SOI + what follows the container segments"""
    data = open(SAMPLE, "rb").read()
    xmp_len = (data[4] << 8) | data[5]
    mpf_at = 4 + xmp_len
    mpf_len = (data[mpf_at + 2] << 8) | data[mpf_at + 3]
    primary = b"\xff\xd8" + data[mpf_at + 2 + mpf_len:42326]
    g = data[42326:]
    gx = (g[4] << 8) | g[5]
    return data, primary, b"\xff\xd8" + g[4 + gx:]


def _append(lib, api, primary, gainmap, md, exif=None, icc=None, cap=None):
    p, g = np.frombuffer(primary, np.uint8), np.frombuffer(gainmap, np.uint8)
    e = None if exif is None else np.frombuffer(exif, np.uint8)
    i = None if icc is None else np.frombuffer(icc, np.uint8)
    out = np.zeros(len(primary) + len(gainmap) + 8192 if cap is None else cap, np.uint8)
    n = C.c_size_t()
    rc = lib.uhdr_hip_jpegr_append_gainmap(C.c_void_p(p.ctypes.data), p.size, C.c_void_p(g.ctypes.data), g.size, None if e is None else C.c_void_p(e.ctypes.data),
                                           0 if e is None else e.size, None if i is None else C.c_void_p(i.ctypes.data), 0 if i is None else i.size, C.byref(md),
                                           C.c_void_p(out.ctypes.data) if out.size else None, out.size, C.byref(n))
    return rc, out[:n.value].tobytes() if rc == 0 else n.value


SAMPLE_MD = dict(version="1.0", max=np.float32(10.0), min=np.float32(1.0), gamma=np.float32(1.0), off_sdr=np.float32(0.0), off_hdr=np.float32(0.0),
                 capmin=np.float32(1.0), capmax=np.float32(10.0))
EXIF = b"Exif\0\0MM\0*\0\0\0\x08\0\x01\x01\x12\0\x03\0\0\0\x01\0\x06\0\0\0\0\0\0"


def test_reassembling_the_sample_file_reproduces_it_byte_for_byte(orc):
    """appendGainMap's XMP and MPF writers (product library, host code -- no GPU involved -- and the Python restatement)
    against the reference's own output: tests/data/sample_jpegr.jpeg"""
    from libultrahdr_dev_amd import api
    from oracle import jpegr_oracle as J
    data, primary, gainmap = _sample_streams()
    assert J.append_gainmap(primary, gainmap, SAMPLE_MD) == data
    lib = api.load()
    hmd = api.metadata(10.0)
    assert _append(lib, api, primary, gainmap, hmd) == (0, data)
    assert _append(lib, api, primary, gainmap, hmd, cap=0) == (api.ERROR_INSUFFICIENT_RESOURCE, len(data))
    assert _append(lib, api, primary, gainmap, api.metadata(10.0, version=b"1.1"))[0] == api.ERROR_BAD_METADATA
    n = C.c_size_t()
    # the ICC segment inside the sample's primary JPEG is writeIccProfile(TF_SRGB, BT709)
    i = data.find(b"ICC_PROFILE\0")
    seg_len = (data[i - 2] << 8) | data[i - 1]
    want = data[i:i - 2 + seg_len]
    assert J.icc_profile_srgb_transfer(orc.CG_BT709) == want
    icc = np.zeros(4096, np.uint8)
    assert lib.uhdr_hip_icc_profile(api.TF_SRGB, api.CG_BT709, C.c_void_p(icc.ctypes.data), icc.size, C.byref(n)) == 0 and icc[:n.value].tobytes() == want
    for gamut in (api.CG_P3, api.CG_BT2100):       # the other two gamuts: product == restatement, and the decoder's gamut reader recognises them
        assert lib.uhdr_hip_icc_profile(api.TF_SRGB, gamut, C.c_void_p(icc.ctypes.data), icc.size, C.byref(n)) == 0
        assert icc[:n.value].tobytes() == J.icc_profile_srgb_transfer(gamut) and J.gamut_from_icc(icc[:n.value].tobytes()) == gamut
    assert lib.uhdr_hip_icc_profile(api.TF_HLG, api.CG_BT709, C.c_void_p(icc.ctypes.data), icc.size, C.byref(n)) == api.ERROR_UNSUPPORTED_FEATURE


def test_write_xmp_then_read_and_icc_write_then_read():
    """the reference's JpegRTest.writeXmpThenRead (tests/jpegr_test.cpp:1401-1432) and IccHelperTest.iccWriteThenRead / iccEndianness
    (tests/icchelper_test.cpp:41-75) against the product library's host code"""
    from libultrahdr_dev_amd import api
    from oracle import jpegr_oracle as J
    lib = api.load()
    _, primary, gainmap = _sample_streams()
    want = api.Metadata(b"1.0", 1.25, 0.75, 1.0, 0.0, 0.0, 1.0, 1.25)
    rc, blob = _append(lib, api, primary, gainmap, want)
    assert rc == 0
    b, got = np.frombuffer(blob, np.uint8), api.Metadata()
    assert lib.uhdr_hip_jpegr_metadata(C.c_void_p(b.ctypes.data), b.size, C.byref(got)) == 0
    for f in ("maxContentBoost", "minContentBoost", "gamma", "offsetSdr", "offsetHdr", "hdrCapacityMin", "hdrCapacityMax"):
        assert abs(getattr(got, f) - getattr(want, f)) <= 4 * np.finfo(np.float32).eps * max(1.0, abs(getattr(want, f))), f     # EXPECT_FLOAT_EQ: 4 ULP
    assert got.version == b"1.0"
    md = J.metadata_from_xmp(J.app_segment(blob[J.find_images(blob)[1][0]:], 0xE1, J.XMP_NS))       # the restatement reads the same values
    assert (md["max"], md["min"]) == (np.float32(got.maxContentBoost), np.float32(got.minContentBoost))
    p = np.frombuffer(primary, np.uint8)
    assert lib.uhdr_hip_jpegr_metadata(C.c_void_p(p.ctypes.data), p.size, C.byref(got)) == api.ERROR_GAIN_MAP_IMAGE_NOT_FOUND
    noxmp = np.frombuffer(primary + gainmap, np.uint8)
    assert lib.uhdr_hip_jpegr_metadata(C.c_void_p(noxmp.ctypes.data), noxmp.size, C.byref(got)) == api.ERROR_METADATA_ERROR
    icc, n = np.zeros(4096, np.uint8), C.c_size_t()
    for gamut in (api.CG_BT709, api.CG_P3, api.CG_BT2100):
        assert lib.uhdr_hip_icc_profile(api.TF_SRGB, gamut, C.c_void_p(icc.ctypes.data), icc.size, C.byref(n)) == 0 and n.value > 14
        prof = icc[:n.value].tobytes()
        assert J.gamut_from_icc(prof) == gamut                                       # iccWriteThenRead
        assert struct.unpack(">I", prof[14:18])[0] == len(prof) - 14                 # iccEndianness: big-endian size field == profile size
    assert lib.uhdr_hip_icc_profile(api.TF_SRGB, 7, C.c_void_p(icc.ctypes.data), icc.size, C.byref(n)) == api.ERROR_INVALID_COLORGAMUT


def test_metadata_packet_in_app2_and_behind_fill_bytes():
    """the reference's decoder takes the first XMP packet of either APP1 or APP2 from libjpeg's marker list
    (jpegdecoderhelper.cpp:221-249), and libjpeg skips 0xFF fill bytes in front of a marker: a file whose gain-map XMP sits in
    an APP2 segment behind a fill byte reads like the original (the same rule uhdr_hip_jpegr_info reports offsets by)"""
    from libultrahdr_dev_amd import api
    from oracle import jpegr_oracle as J
    lib = api.load()
    _, primary, gainmap = _sample_streams()
    want = api.Metadata(b"1.0", 2.5, 1.0, 1.0, 0.0, 0.0, 1.0, 2.5)
    rc, blob = _append(lib, api, primary, gainmap, want)
    assert rc == 0
    second = J.find_images(blob)[1][0]
    at = blob.index(b"\xff\xe1", second)
    assert blob[at + 4:at + 4 + len(J.XMP_NS)] == J.XMP_NS
    moved = blob[:at] + b"\xff\xff\xe2" + blob[at + 2:]            # fill byte, then the same segment as APP2
    seen = []
    for data in (blob, moved):
        b, got = np.frombuffer(data, np.uint8), api.Metadata()
        assert lib.uhdr_hip_jpegr_metadata(C.c_void_p(b.ctypes.data), b.size, C.byref(got)) == 0
        assert abs(got.maxContentBoost - 2.5) < 1e-5 and got.minContentBoost == 1.0 and got.version == b"1.0"   # (the packet stores log2 with %g)
        seen.append((got.maxContentBoost, got.minContentBoost, got.hdrCapacityMax))
    assert seen[0] == seen[1]


def test_append_gainmap_exif_and_icc_handling(orc):
    """appendGainMap's EXIF / ICC arguments and the EXIF segment it lifts out of the primary JPEG (jpegr.cpp:1003-1071); host code,
    product == restatement.  What the restatement rests on: the segment order documented at jpegr.cpp:917-949 and the sample pin above."""
    from libultrahdr_dev_amd import api
    from oracle import jpegr_oracle as J
    lib = api.load()
    data, primary, gainmap = _sample_streams()
    hmd = api.metadata(10.0)
    icc = J.icc_profile_srgb_transfer(orc.CG_P3)
    seg = lambda m, b: bytes((0xFF, m)) + struct.pack(">H", len(b) + 2) + b
    with_exif = primary[:2] + seg(0xE1, EXIF) + primary[2:]                            # EXIF right after SOI
    after_app0 = primary[:2] + seg(0xE0, b"JFIF\0\1\1\0\0\1\0\1\0\0") + seg(0xE1, EXIF) + primary[2:]
    after_app2 = primary[:2] + seg(0xE2, b"X" * 10) + seg(0xE1, EXIF) + primary[2:]     # an unsaved segment in front: the reference's position is off
    for prim, ex, ic in ((primary, EXIF, None), (primary, None, icc), (primary, EXIF, icc), (with_exif, None, None), (after_app0, None, icc),
                         (after_app2, None, None)):
        want = J.append_gainmap(prim, gainmap, SAMPLE_MD, exif=ex, icc=ic)
        assert isinstance(want, bytes) and _append(lib, api, prim, gainmap, hmd, ex, ic) == (0, want)
    plain = J.append_gainmap(primary, gainmap, SAMPLE_MD, exif=EXIF)
    assert J.append_gainmap(with_exif, gainmap, SAMPLE_MD) == plain                     # lifted EXIF lands where an external one would
    assert plain[:4] == b"\xff\xd8\xff\xe1" and plain[6:12] == b"Exif\0\0"
    assert _append(lib, api, with_exif, gainmap, hmd, EXIF)[0] == -20007 == J.append_gainmap(with_exif, gainmap, SAMPLE_MD, exif=EXIF)
    assert _append(lib, api, primary[:200], gainmap, hmd)[0] == api.ERROR_DECODE_ERROR == J.append_gainmap(primary[:200], gainmap, SAMPLE_MD)
    assert _append(lib, api, b"\0\0" + primary[2:], gainmap, hmd)[0] == api.ERROR_DECODE_ERROR
    # files with EXIF + ICC up front still split and parse (decode side of the same container)
    both = J.append_gainmap(primary, gainmap, SAMPLE_MD, exif=EXIF, icc=icc)
    imgs = J.find_images(both)
    assert len(imgs) == 2 and both[imgs[1][0]:imgs[1][0] + imgs[1][1]][-len(gainmap) + 2:] == gainmap[2:]


def test_api4_and_info_on_host(orc):
    """encodeJPEGR API-4 (jpegr.cpp:502-560) and getJPEGRInfo (:633-653): host-only entry points of the product library"""
    from libultrahdr_dev_amd import api
    from oracle import jpegr_oracle as J
    lib = api.load()
    data, primary, gainmap = _sample_streams()
    hmd = api.metadata(10.0)

    def api4(prim, gamut, md=hmd, cap=1 << 17):
        p, g = np.frombuffer(prim, np.uint8), np.frombuffer(gainmap, np.uint8)
        out, n = np.zeros(cap, np.uint8), C.c_size_t()
        rc = lib.uhdr_hip_jpegr_encode_api4(C.c_void_p(p.ctypes.data), p.size, gamut, C.c_void_p(g.ctypes.data), g.size, None if md is None else C.byref(md),
                                            C.c_void_p(out.ctypes.data), out.size, C.byref(n))
        return rc, out[:n.value].tobytes() if rc == 0 else None
    assert api4(primary, api.CG_UNSPECIFIED) == (0, data)                       # the sample's primary carries an ICC profile: nothing added
    i = primary.find(b"ICC_PROFILE\0")
    seg_len = (primary[i - 2] << 8) | primary[i - 1]
    no_icc = primary[:i - 4] + primary[i - 2 + seg_len:]
    assert J.app_segment(no_icc, 0xE2, J.ICC_ID) is None
    for gamut in (api.CG_BT709, api.CG_P3, api.CG_BT2100):
        want = J.encode_api4(no_icc, gamut, gainmap, SAMPLE_MD)
        assert api4(no_icc, gamut) == (0, want) and J.gamut_from_icc(J.app_segment(want, 0xE2, J.ICC_ID)) == gamut
    assert api4(no_icc, api.CG_UNSPECIFIED)[0] == api.ERROR_INVALID_COLORGAMUT == J.encode_api4(no_icc, -1, gainmap, SAMPLE_MD)
    assert api4(primary[:100], api.CG_BT709)[0] == api.ERROR_DECODE_ERROR == J.encode_api4(primary[:100], 0, gainmap, SAMPLE_MD)
    assert api4(primary, api.CG_BT709, md=None)[0] == api.ERROR_BAD_PTR
    assert api4(primary, api.CG_BT709, cap=100)[0] == api.ERROR_INSUFFICIENT_RESOURCE

    def info(blob, want_gainmap=True):
        b = np.frombuffer(blob, np.uint8)
        a, g = api.JpegInfo(), api.JpegInfo()
        rc = lib.uhdr_hip_jpegr_info(C.c_void_p(b.ctypes.data), b.size, C.byref(a), C.byref(g) if want_gainmap else None)
        conv = lambda x: dict(offset=x.offset, size=x.size, width=x.width, height=x.height, icc=(x.icc_offset, x.icc_size), exif=(x.exif_offset, x.exif_size),
                              xmp=(x.xmp_offset, x.xmp_size))
        return rc if rc != 0 else [conv(a), conv(g)][:2 if want_gainmap else 1]
    got = info(data)
    assert got == J.info(data)
    assert (got[0]["width"], got[0]["height"], got[0]["offset"], got[0]["size"]) == (1280, 720, 0, 42326) and got[1]["size"] == 3727
    assert (got[1]["width"], got[1]["height"]) == (320, 180)
    assert data[got[0]["icc"][0]:][:12] == b"ICC_PROFILE\0" and got[0]["icc"][1] == 602 and got[0]["exif"] == (0, 0)
    assert data[got[1]["xmp"][0]:][:29] == J.XMP_NS and b"hdrgm:GainMapMax" in data[got[1]["xmp"][0]:got[1]["xmp"][0] + got[1]["xmp"][1]]
    assert info(data, False) == J.info(data)[:1]
    both = J.append_gainmap(primary, gainmap, SAMPLE_MD, exif=EXIF)
    assert info(both) == J.info(both) and info(both)[0]["exif"][1] == len(EXIF)
    assert info(primary) == -20003 == J.info(primary) and info(b"\0" * 64) == -20006 == J.info(b"\0" * 64)


class _Enc:
    """calls the encodeJPEGR entry points with tightly packed numpy inputs living on the host or on the device"""

    def __init__(self, hip, device):
        from tests.gpu_util import stream_ptr, to_dev
        self.hip, self.lib, self.device = hip, hip.load(), device
        self.ms, self.stream = (hip.MEM_DEVICE, stream_ptr()) if device else (hip.MEM_HOST, None)
        self._keep, self._to_dev = [], to_dev

    def _img(self, maker, arr, w, h, gamut, defaults=True, **kw):
        if self.device:
            d = self._to_dev(arr)
            self._keep.append(d)
            return maker(d.data_ptr(), w, h, gamut, **kw)
        im = maker(arr.ctypes.data, w, h, gamut, **kw)
        if defaults and not kw:                      # the reference's "0 / nullptr means packed" defaults (jpegr.cpp:261-275)
            im.chroma_data, im.luma_stride, im.chroma_stride = None, 0, 0
        return im

    def p010(self, arr, w, h, gamut, **kw):
        return self._img(self.hip.p010_image, arr, w, h, gamut, **kw)

    def yuv(self, arr, w, h, gamut, **kw):
        return self._img(self.hip.yuv420_image, arr, w, h, gamut, **kw)

    def run(self, name, *args, cap=1 << 22):
        out, n = np.zeros(cap, np.uint8), C.c_size_t()
        conv = []
        for a in args:
            if isinstance(a, bytes):
                b = np.frombuffer(a, np.uint8)
                self._keep.append(b)
                conv += [C.c_void_p(b.ctypes.data), b.size]
            elif a is None:
                conv += [None, 0]
            elif isinstance(a, (self.hip.Image, self.hip.Metadata)):
                conv.append(C.byref(a))
            else:
                conv.append(a)
        tail = [C.c_void_p(out.ctypes.data), out.size, C.byref(n)] + ([] if name == "api4" else [self.ms, self.stream])
        rc = getattr(self.lib, "uhdr_hip_jpegr_encode_" + name)(*conv, *tail)
        return rc, (out[:n.value].tobytes() if rc == 0 else n.value)


ENC_CASES = (((640, 480), 0, 1, 95), ((200, 120), 1, 2, 80), ((72, 40), 2, 0, 100))      # (w, h), SDR gamut, hdr_tf, quality


@pytest.mark.gpu
@pytest.mark.parametrize("device", [True, False])
def test_gpu_encode_api0_api1_equal_the_restatement_and_round_trip(hip, orc, device):
    """encodeJPEGR API-0 / API-1 on the device (BASELINE configs[0]: 640x480 P010 + YUV420) == the CPU restatement, byte for byte; the
    files then decode through uhdr_hip_jpegr_decode"""
    from oracle import jpegr_oracle as J
    from tests.test_gpu_parity import smooth_frame
    e = _Enc(hip, device)
    for (w, h), sg, tf, q in ENC_CASES:
        p010, yuv = smooth_frame(w, h, w)
        for exif in (None, EXIF):
            want = J.encode_api1(p010, yuv, w, h, sg, hip.CG_BT2100, tf, q, exif=exif)
            rc, got = e.run("api1", e.p010(p010, w, h, hip.CG_BT2100), e.yuv(yuv, w, h, sg), tf, q, exif)
            assert rc == 0 and got == want, ("api1", w, h, sg, tf)
        st, ref, ow, oh, gamut, md = J.decode(got, orc.OUT_HDR_HLG, FLT_MAX)
        rc, dec, dest, _ = _gpu_decode(e.lib, hip, got, hip.OUTPUT_HDR_HLG, FLT_MAX, hip.APPLY_EXACT, device)
        assert rc == st == 0 and (dest.width, dest.height, dest.colorGamut) == (w, h, sg) and np.array_equal(dec, ref)
        want = J.encode_api0(p010, w, h, sg, tf, q, exif=EXIF)                # the HDR gamut is also the SDR gamut on this path
        rc, got = e.run("api0", e.p010(p010, w, h, sg), tf, q, EXIF)
        assert rc == 0 and got == want, ("api0", w, h, sg, tf)
        assert e.run("api0", e.p010(p010, w, h, sg), tf, q, None, cap=64) == (hip.ERROR_INSUFFICIENT_RESOURCE, len(J.encode_api0(p010, w, h, sg, tf, q)))
    # strided inputs with separate chroma planes: same file as the packed ones
    w, h, ls = 104, 56, 128
    p010, yuv = smooth_frame(w, h, 5)
    sp = np.zeros(ls * h * 3 // 2, np.uint16)
    sp[:ls * h].reshape(h, ls)[:, :w] = p010[:w * h].reshape(h, w)
    sp[ls * h:].reshape(h // 2, ls)[:, :w] = p010[w * h:].reshape(h // 2, w)
    sy = np.full(ls * h * 3 // 2, 0x55, np.uint8)
    sy[:ls * h].reshape(h, ls)[:, :w] = yuv[:w * h].reshape(h, w)
    cw, ch, cs = w // 2, h // 2, ls // 2
    sy[ls * h:ls * h + cs * ch].reshape(ch, cs)[:, :cw] = yuv[w * h:w * h + cw * ch].reshape(ch, cw)
    sy[ls * h + cs * ch:].reshape(ch, cs)[:, :cw] = yuv[w * h + cw * ch:].reshape(ch, cw)
    want = J.encode_api1(p010, yuv, w, h, hip.CG_BT709, hip.CG_BT2100, hip.TF_HLG, 90)
    rc, got = e.run("api1", e.p010(sp, w, h, hip.CG_BT2100, luma_stride=ls), e.yuv(sy, w, h, hip.CG_BT709, luma_stride=ls), hip.TF_HLG, 90, None)
    assert rc == 0 and got == want


@pytest.mark.gpu
@pytest.mark.parametrize("device", [True, False])
def test_gpu_encode_api2_api3_apix_equal_the_restatement(hip, orc, device):
    """encodeJPEGR API-2 / API-3 (gain map for a given SDR JPEG; API-3 decodes it on the device and treats the planes as BT.601) and
    API-x (ready gain map) == the CPU restatement, byte for byte"""
    from oracle import jpegr_oracle as J
    from tests.test_gpu_parity import smooth_frame
    e = _Enc(hip, device)
    for (w, h), sg, tf, q in ENC_CASES:
        p010, yuv = smooth_frame(w, h, w + 1)
        for with_icc in (True, False):
            sdr_jpeg = orc.jpeg_encode("orc", yuv[:w * h], yuv[w * h:], w, h, q, icc=J.icc_profile_srgb_transfer(sg) if with_icc else None)
            want = J.encode_api2(p010, yuv, w, h, sg, hip.CG_BT2100, sdr_jpeg, sg, tf)
            rc, got = e.run("api2", e.p010(p010, w, h, hip.CG_BT2100), e.yuv(yuv, w, h, sg), sdr_jpeg, sg, tf)
            assert rc == 0 and got == want, ("api2", w, h, with_icc)
            for cfg in (sg, hip.CG_UNSPECIFIED):
                want = J.encode_api3(p010, w, h, hip.CG_BT2100, sdr_jpeg, cfg, tf)
                rc, got = e.run("api3", e.p010(p010, w, h, hip.CG_BT2100), sdr_jpeg, cfg, tf)
                if isinstance(want, bytes):
                    assert rc == 0 and got == want, ("api3", w, h, with_icc, cfg)
                else:
                    assert rc == want == hip.ERROR_INVALID_COLORGAMUT and not with_icc
        # the SDR JPEG's ICC gamut disagreeing with the configured one (jpegr.cpp:470-478), and a size mismatch (:496-499)
        sdr_jpeg = orc.jpeg_encode("orc", yuv[:w * h], yuv[w * h:], w, h, q, icc=J.icc_profile_srgb_transfer(sg))
        assert e.run("api3", e.p010(p010, w, h, hip.CG_BT2100), sdr_jpeg, (sg + 1) % 3, tf)[0] == hip.ERROR_INVALID_COLORGAMUT
        p2, _ = smooth_frame(w + 8, h, 3)
        assert e.run("api3", e.p010(p2, w + 8, h, hip.CG_BT2100), sdr_jpeg, sg, tf)[0] == hip.ERROR_RESOLUTION_MISMATCH
        assert e.run("api3", e.p010(p010, w, h, hip.CG_BT2100), sdr_jpeg[:len(sdr_jpeg) // 2], sg, tf)[0] == hip.ERROR_DECODE_ERROR
        # API-x
        gmap = (np.arange((w // 4) * (h // 4), dtype=np.uint32) * 7 % 251).astype(np.uint8).reshape(h // 4, w // 4)
        md = dict(version="1.0", max=np.float32(6.5), min=np.float32(0.5), gamma=np.float32(1.0), off_sdr=np.float32(0.015625), off_hdr=np.float32(0.015625),
                  capmin=np.float32(1.0), capmax=np.float32(6.5))
        hmd = hip.metadata(6.5, 0.5)
        hmd.offsetSdr = hmd.offsetHdr = 0.015625
        hmd.hdrCapacityMin = 1.0
        want = J.encode_apix(yuv, w, h, sg, gmap, md, q, exif=EXIF)
        g = e._img(lambda ptr, gw, gh, _g: hip.mono_image(ptr, gw, gh), gmap.reshape(-1), w // 4, h // 4, -1, defaults=False)
        rc, got = e.run("apix", e.yuv(yuv, w, h, sg), g, hmd, q, EXIF)
        assert rc == 0 and got == want, ("apix", w, h)
        assert J.metadata_from_xmp(J.app_segment(got[J.find_images(got)[1][0]:], 0xE1, J.XMP_NS))["min"] == np.float32(0.5)


@pytest.mark.gpu
def test_gpu_encode_status_codes_in_the_reference_order(hip):
    """areInputArgumentsValid (jpegr.cpp:75-183) and the per-overload checks"""
    from tests.test_gpu_parity import smooth_frame
    e = _Enc(hip, False)
    p010, yuv = smooth_frame(64, 48, 1)
    P = lambda w=64, h=48, g=hip.CG_BT2100, **kw: e.p010(p010, w, h, g, **kw)
    Y = lambda w=64, h=48, g=hip.CG_BT709, **kw: e.yuv(yuv, w, h, g, **kw)
    api1 = lambda p, y, tf=hip.TF_HLG, q=90: e.run("api1", p, y, tf, q, None)[0]
    assert api1(P(), Y()) == 0
    assert e.lib.uhdr_hip_jpegr_encode_api1(C.byref(P()), None, 1, 90, None, 0, None, 0, None, 0, None) == hip.ERROR_BAD_PTR
    assert api1(P(), Y(), q=101) == hip.ERROR_INVALID_QUALITY_FACTOR == api1(P(63), Y(), q=-1)          # quality before anything about the images
    assert api1(P(63), Y()) == hip.ERROR_UNSUPPORTED_WIDTH_HEIGHT == api1(P(64, 47), Y()) == api1(P(4, 4), Y()) == api1(P(8194, 48), Y())
    assert api1(P(g=hip.CG_UNSPECIFIED), Y()) == hip.ERROR_INVALID_COLORGAMUT == api1(P(), Y(g=3))
    assert api1(P(luma_stride=32), Y()) == hip.ERROR_INVALID_STRIDE == api1(P(), Y(luma_stride=32)) == api1(P(), Y(chroma_stride=16))
    assert api1(P(), Y(), tf=hip.TF_SRGB) == hip.ERROR_INVALID_TRANS_FUNC == api1(P(), Y(), tf=-1) == api1(P(), Y(32), tf=4)   # before the YUV checks
    assert api1(P(), Y(32)) == hip.ERROR_RESOLUTION_MISMATCH == api1(P(), Y(64, 32))
    assert e.run("api0", P(), hip.TF_PQ, 101, None)[0] == hip.ERROR_INVALID_QUALITY_FACTOR and e.run("api0", P(63), hip.TF_PQ, 90, None)[0] == -10002
    assert e.lib.uhdr_hip_jpegr_encode_api0(C.byref(P()), 1, 90, None, 12, C.c_void_p(p010.ctypes.data), 64, C.byref(C.c_size_t()), 0, None) == hip.ERROR_BAD_PTR
    assert e.run("api2", P(), Y(), b"\xff\xd8\xff\xd9", 0, hip.TF_HLG)[0] == hip.ERROR_DECODE_ERROR
    assert e.run("api3", P(), b"\xff\xd8\xff\xd9", 0, hip.TF_HLG)[0] == hip.ERROR_DECODE_ERROR


def test_host_parsers_survive_mutated_files_under_asan(orc, tmp_path):
    """mutation fuzzing of every host-side parser that sees untrusted bytes (container scan, XMP, ICC, EXIF lifting, the JPEG header parser
    that sizes the decoder's device buffers), built with AddressSanitizer + UBSan on the CPU -- tests/cpp/fuzz_host_parsers.cpp"""
    import subprocess
    from oracle import jpegr_oracle as J
    rng = np.random.RandomState(11)
    w, h = 64, 48
    y = rng.randint(0, 256, w * h * 3 // 2).astype(np.uint8)
    primary = orc.jpeg_encode("orc", y[:w * h], y[w * h:], w, h, 90, icc=J.icc_profile_srgb_transfer(orc.CG_P3))
    gray = orc.jpeg_encode("orc", rng.randint(0, 256, 16 * 12).astype(np.uint8), None, 16, 12, 85)
    small = J.append_gainmap(primary, gray, SAMPLE_MD, exif=EXIF)
    seeds = []
    for name, blob in (("small.jpgr", small), ("primary.jpg", primary), ("gray.jpg", gray)):
        (tmp_path / name).write_bytes(blob)
        seeds.append(str(tmp_path / name))
    try:      # a file with restart intervals for the marker walk of the header parser, when Pillow can write one here
        import io
        from PIL import Image
        b = io.BytesIO()
        Image.fromarray(y[:w * h].reshape(h, w), mode="L").save(b, "JPEG", quality=80, restart_marker_blocks=2)
        (tmp_path / "rst.jpg").write_bytes(b.getvalue())
        seeds.append(str(tmp_path / "rst.jpg"))
        # progressive files: every scan is entropy-decoded by host code (uhdr_jpeg_prog.cpp) -- colour with restart intervals, gray
        rgb = np.stack([y[:w * h].reshape(h, w)] * 3, axis=-1)
        for name, img, kw in (("prog.jpg", Image.fromarray(rgb, mode="RGB"), dict(subsampling="4:2:0", restart_marker_blocks=1)),
                              ("prog_gray.jpg", Image.fromarray(y[:w * h].reshape(h, w), mode="L"), dict(optimize=True))):
            b = io.BytesIO()
            img.save(b, "JPEG", quality=85, progressive=True, **kw)
            (tmp_path / name).write_bytes(b.getvalue())
            seeds.append(str(tmp_path / name))
    except Exception:
        pass
    exe = str(tmp_path / "fuzz")
    csrc = os.path.join(ROOT, "libultrahdr_dev_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-D__HIP_PLATFORM_AMD__",
                           "-I/opt/rocm/include", os.path.join(ROOT, "tests", "cpp", "fuzz_host_parsers.cpp"), os.path.join(csrc, "uhdr_jpegr.cpp"),
                           os.path.join(csrc, "uhdr_jpeg_hdr.cpp"), os.path.join(csrc, "uhdr_jpeg_prog.cpp"), "-o", exe])
    r = subprocess.run([exe, SAMPLE] + seeds + ["60000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "fuzz ok" in r.stdout, (r.stdout + r.stderr)[-3000:]


def _strided_p010(p010, w, h, ls, cs, separate):
    """the fixture re-laid with luma stride ls and chroma stride cs (elements); separate: chroma in its own buffer (the reference's
    UhdrUnCompressedStructWrapper::setChromaMode(false), tests/jpegr_test.cpp:190-230)"""
    y = np.full(ls * h, 0xABCD, np.uint16)
    y.reshape(h, ls)[:, :w] = p010[:w * h].reshape(h, w)
    c = np.full(cs * (h // 2), 0xABCD, np.uint16)
    c.reshape(h // 2, cs)[:, :w] = p010[w * h:].reshape(h // 2, w)
    return (y, c) if separate else (np.concatenate([y, c]), None)


def _strided_yuv(yuv, w, h, ls, cs, separate):
    y = np.full(ls * h, 0x5A, np.uint8)
    y.reshape(h, ls)[:, :w] = yuv[:w * h].reshape(h, w)
    cw, ch = w // 2, h // 2
    c = np.full(cs * ch * 2, 0x5A, np.uint8)
    c[:cs * ch].reshape(ch, cs)[:, :cw] = yuv[w * h:w * h + cw * ch].reshape(ch, cw)
    c[cs * ch:].reshape(ch, cs)[:, :cw] = yuv[w * h + cw * ch:].reshape(ch, cw)
    return (y, c) if separate else (np.concatenate([y, c]), None)


@pytest.mark.gpu
@pytest.mark.parametrize("p010_gamut", [0, 1, 2])
@pytest.mark.parametrize("sdr_gamut", [0, 1, 2])
def test_gpu_encode_is_invariant_to_strides_and_chroma_pointers(hip, orc, p010_gamut, sdr_gamut):
    """the reference's JpegRAPIEncodeAndDecodeTest (tests/jpegr_test.cpp:1434-1801): the 1280x720 fixture pair over the 3x3 gamut
    combinations through API-0 and API-1, re-laid with the luma / chroma strides and separate chroma buffers that test uses; every
    variant must give the file of the packed layout (the reference's memcmp), and that file is the CPU restatement's"""
    from oracle import jpegr_oracle as J
    w, h, q = 1280, 720, 90
    p010 = np.fromfile(os.path.join(GOLDEN, "raw_p010_image.p010"), np.uint16)
    yuv = np.fromfile(os.path.join(GOLDEN, "raw_yuv420_image.yuv420"), np.uint8)
    lib = hip.load()

    def p_img(ls, cs, separate, give_cs=True):
        a, c = _strided_p010(p010, w, h, ls or w, (cs or ls or w) if separate else (ls or w), separate)
        im = hip.p010_image(a.ctypes.data, w, h, p010_gamut)
        im.luma_stride, im.chroma_stride = ls, cs if give_cs else 0
        im.chroma_data = c.ctypes.data if separate else None
        return im, (a, c)

    def y_img(ls, cs, separate):
        a, c = _strided_yuv(yuv, w, h, ls or w, (cs or (ls or w) // 2) if separate else (ls or w) // 2, separate)
        im = hip.yuv420_image(a.ctypes.data, w, h, sdr_gamut)
        im.luma_stride, im.chroma_stride = ls, cs
        im.chroma_data = c.ctypes.data if separate else None
        return im, (a, c)

    out, n = np.zeros(w * h * 3, np.uint8), C.c_size_t()

    def api0(pi):
        rc = lib.uhdr_hip_jpegr_encode_api0(C.byref(pi[0]), hip.TF_HLG, q, None, 0, C.c_void_p(out.ctypes.data), out.size, C.byref(n), hip.MEM_HOST, None)
        assert rc == 0
        return out[:n.value].tobytes()

    def api1(pi, yi):
        rc = lib.uhdr_hip_jpegr_encode_api1(C.byref(pi[0]), C.byref(yi[0]), hip.TF_HLG, q, None, 0, C.c_void_p(out.ctypes.data), out.size, C.byref(n), hip.MEM_HOST, None)
        assert rc == 0
        return out[:n.value].tobytes()

    # the P010 layouts of EncodeAPI0AndDecodeTest (:1485-1582): luma stride; luma + chroma strides with a chroma pointer; chroma stride
    # alone with a chroma pointer; both strides but no chroma pointer (the chroma stride is then ignored, jpegr.cpp:265-270)
    p_layouts = [(0, 0, False), (w + 18, 0, False), (w + 18, w + 28, True), (0, w + 34, True), (w, w + 38, False)]
    base0 = api0(p_img(*p_layouts[0]))
    assert base0 == J.encode_api0(p010, w, h, p010_gamut, hip.TF_HLG, q)
    for lay in p_layouts[1:]:
        assert api0(p_img(*lay)) == base0, lay
    # EncodeAPI1AndDecodeTest (:1611-1801): the same on either input
    y_layouts = [(0, 0, False), (w + 14, 0, False), (w + 46, w // 2 + 34, True), (0, w // 2 + 38, True), (w + 26, 0, True)]
    base1 = api1(p_img(*p_layouts[0]), y_img(*y_layouts[0]))
    assert base1 == J.encode_api1(p010, yuv, w, h, sdr_gamut, p010_gamut, hip.TF_HLG, q)
    for lay in p_layouts[1:]:
        assert api1(p_img(*lay), y_img(*y_layouts[0])) == base1, ("p010", lay)
    for lay in y_layouts[1:]:
        if lay[2] and lay[1] == 0:
            continue                   # a chroma pointer with chroma_stride 0 fails validation (:150-155): not a layout
        assert api1(p_img(*p_layouts[0]), y_img(*lay)) == base1, ("yuv", lay)
    # EncodeAPI2AndDecodeTest / EncodeAPI3AndDecodeTest (:1815-2248): the gain map for a given SDR JPEG, same layouts
    sdr_jpeg = np.frombuffer(orc.jpeg_encode("orc", yuv[:w * h], yuv[w * h:], w, h, q, icc=J.icc_profile_srgb_transfer(sdr_gamut)), np.uint8)

    def api2(pi, yi):
        rc = lib.uhdr_hip_jpegr_encode_api2(C.byref(pi[0]), C.byref(yi[0]), C.c_void_p(sdr_jpeg.ctypes.data), sdr_jpeg.size, sdr_gamut, hip.TF_HLG,
                                            C.c_void_p(out.ctypes.data), out.size, C.byref(n), hip.MEM_HOST, None)
        assert rc == 0
        return out[:n.value].tobytes()

    def api3(pi):
        rc = lib.uhdr_hip_jpegr_encode_api3(C.byref(pi[0]), C.c_void_p(sdr_jpeg.ctypes.data), sdr_jpeg.size, sdr_gamut, hip.TF_HLG, C.c_void_p(out.ctypes.data), out.size,
                                            C.byref(n), hip.MEM_HOST, None)
        assert rc == 0
        return out[:n.value].tobytes()
    base2, base3 = api2(p_img(*p_layouts[0]), y_img(*y_layouts[0])), api3(p_img(*p_layouts[0]))
    assert base2 == J.encode_api2(p010, yuv, w, h, sdr_gamut, p010_gamut, sdr_jpeg.tobytes(), sdr_gamut, hip.TF_HLG)
    assert base3 == J.encode_api3(p010, w, h, p010_gamut, sdr_jpeg.tobytes(), sdr_gamut, hip.TF_HLG)
    for lay in p_layouts[1:]:
        assert api2(p_img(*lay), y_img(*y_layouts[0])) == base2 and api3(p_img(*lay)) == base3, ("p010", lay)
    for lay in y_layouts[1:4]:
        assert api2(p_img(*p_layouts[0]), y_img(*lay)) == base2, ("yuv", lay)


def test_invalid_argument_matrices_need_no_gpu():
    """the reference's EncodeAPI0..4WithInvalidArgs / DecodeAPIWithInvalidArgs (tests/jpegr_test.cpp:386-1399; it only asserts != NO_ERROR,
    here the status each check returns, read off areInputArgumentsValid jpegr.cpp:75-183 and the overloads).  All of these are
    decided on the host before the device is looked at, so they run in the CPU suite."""
    from libultrahdr_dev_amd import api
    lib = api.load()
    W, H = 32, 32
    p010 = np.zeros(W * H * 3 // 2 + 4096, np.uint16)
    yuv = np.zeros(W * H * 3 // 2 + 4096, np.uint8)
    out, n = np.zeros(1 << 16, np.uint8), C.c_size_t()
    O = (C.c_void_p(out.ctypes.data), out.size, C.byref(n))

    def P(w=W, h=H, g=api.CG_BT2100, **kw):
        return api.p010_image(p010.ctypes.data, w, h, g, **kw)

    def Y(w=W, h=H, g=api.CG_BT709, **kw):
        return api.yuv420_image(yuv.ctypes.data, w, h, g, **kw)
    jpg = np.frombuffer(b"\xff\xd8\xff\xd9" + bytes(60), np.uint8)
    J = (C.c_void_p(jpg.ctypes.data), jpg.size)
    bad_dims = [P(W - 1), P(W, H - 1), P(0), P(W, 0), P(4, 4), P(8194, H), P(W, 8194)]
    bad_gamut = [P(g=api.CG_UNSPECIFIED), P(g=3)]
    bad_stride = [P(luma_stride=W - 2), P(luma_stride=W + 64, chroma_stride=W - 2, chroma_ptr=p010.ctypes.data)]
    bad_tf = [api.TF_SRGB, -1, 4]

    # API-0 (:386-537)
    a0 = lambda p, tf=api.TF_HLG, q=90, o=O: lib.uhdr_hip_jpegr_encode_api0(None if p is None else C.byref(p), tf, q, None, 0, *o, api.MEM_HOST, None)
    assert a0(P(), q=-1) == a0(P(), q=101) == api.ERROR_INVALID_QUALITY_FACTOR
    nodata_p, nodata_y = P(), Y()
    nodata_p.data = None
    nodata_y.data = None
    assert a0(None) == api.ERROR_BAD_PTR == a0(nodata_p)
    assert all(a0(p) == api.ERROR_UNSUPPORTED_WIDTH_HEIGHT for p in bad_dims)
    assert all(a0(p) == api.ERROR_INVALID_COLORGAMUT for p in bad_gamut)
    assert all(a0(p) == api.ERROR_INVALID_STRIDE for p in bad_stride)
    assert a0(P(), o=(None, 0, C.byref(n))) == api.ERROR_BAD_PTR
    assert all(a0(P(), tf=t) == api.ERROR_INVALID_TRANS_FUNC for t in bad_tf)
    assert lib.uhdr_hip_jpegr_encode_api0(C.byref(P()), api.TF_HLG, 90, None, 8, *O, api.MEM_HOST, None) == api.ERROR_BAD_PTR      # exif struct without data

    # API-2 (:814-1099) and API-3 (:1101-1265)
    a2 = lambda p, y, j=J, tf=api.TF_HLG, o=O: lib.uhdr_hip_jpegr_encode_api2(None if p is None else C.byref(p), None if y is None else C.byref(y), *j, api.CG_BT709,
                                                                               tf, *o, api.MEM_HOST, None)
    a3 = lambda p, j=J, tf=api.TF_HLG, o=O: lib.uhdr_hip_jpegr_encode_api3(None if p is None else C.byref(p), *j, api.CG_BT709, tf, *o, api.MEM_HOST, None)
    assert a2(P(), None) == a2(P(), Y(), j=(None, 0)) == a2(None, Y()) == api.ERROR_BAD_PTR == a3(None) == a3(P(), j=(None, 0))
    for p in bad_dims:
        assert a2(p, Y()) == a3(p) == api.ERROR_UNSUPPORTED_WIDTH_HEIGHT
    for p in bad_gamut:
        assert a2(p, Y()) == a3(p) == api.ERROR_INVALID_COLORGAMUT
    for p in bad_stride:
        assert a2(p, Y()) == a3(p) == api.ERROR_INVALID_STRIDE
    for t in bad_tf:
        assert a2(P(), Y(), tf=t) == a3(P(), tf=t) == api.ERROR_INVALID_TRANS_FUNC
    assert a2(P(), Y(), o=(None, 0, C.byref(n))) == a3(P(), o=(None, 0, C.byref(n))) == api.ERROR_BAD_PTR
    assert a2(P(), nodata_y) == api.ERROR_BAD_PTR
    assert a2(P(), Y(luma_stride=W - 2)) == a2(P(), Y(chroma_stride=W // 2 - 2, chroma_ptr=yuv.ctypes.data)) == api.ERROR_INVALID_STRIDE
    assert a2(P(), Y(W + 2)) == a2(P(), Y(W, H + 2)) == api.ERROR_RESOLUTION_MISMATCH
    assert a2(P(), Y(g=api.CG_UNSPECIFIED)) == a2(P(), Y(g=3)) == api.ERROR_INVALID_COLORGAMUT
    assert a3(P()) == api.ERROR_DECODE_ERROR                                     # "\xff\xd8\xff\xd9": no image in it

    # API-4 (:1267-1360)
    md = api.metadata(4.0)
    a4 = lambda j=J, g=J, m=md, o=O: lib.uhdr_hip_jpegr_encode_api4(*j, api.CG_BT709, *g, None if m is None else C.byref(m), *o)
    assert a4(j=(None, 0)) == a4(g=(None, 0)) == a4(o=(None, 0, C.byref(n))) == api.ERROR_BAD_PTR
    assert a4() == api.ERROR_DECODE_ERROR
    _, primary, gainmap = _sample_streams()
    pj, gj = np.frombuffer(primary, np.uint8), np.frombuffer(gainmap, np.uint8)
    PJ, GJ = (C.c_void_p(pj.ctypes.data), pj.size), (C.c_void_p(gj.ctypes.data), gj.size)
    big = np.zeros(1 << 17, np.uint8)
    BO = (C.c_void_p(big.ctypes.data), big.size, C.byref(n))
    assert a4(PJ, GJ, md, BO) == 0 and a4(PJ, GJ, None, BO) == api.ERROR_BAD_PTR
    for field, val in (("version", b"1.1"), ("maxContentBoost", 0.5), ("hdrCapacityMax", 0.5), ("hdrCapacityMin", 0.5), ("offsetSdr", -1.0), ("offsetHdr", -1.0),
                       ("gamma", 0.0), ("gamma", -1.0)):
        bad = api.metadata(4.0)
        setattr(bad, field, val)
        assert a4(PJ, GJ, bad, BO) == api.ERROR_BAD_METADATA, field

    # decodeJPEGR (:1362-1399): nothing below reaches the device either
    data = np.fromfile(SAMPLE, np.uint8)
    dest, dmd = api.Image(), api.Metadata()
    dec = lambda buf=data, fmt=api.OUTPUT_HDR_LINEAR, boost=api.FLT_MAX, d=dest: lib.uhdr_hip_jpegr_decode(
        None if buf is None else C.c_void_p(buf.ctypes.data), 0 if buf is None else buf.size, fmt, boost, None, 0, None if d is None else C.byref(d), C.byref(dmd),
        api.APPLY_FAST, api.MEM_HOST, None)
    assert dec(buf=None) == dec(d=None) == api.ERROR_BAD_PTR
    assert dec(boost=0.5) == api.ERROR_INVALID_DISPLAY_BOOST
    assert dec(fmt=-1) == dec(fmt=5) == api.ERROR_INVALID_OUTPUT_FORMAT
    assert dec(buf=np.zeros(64, np.uint8)) == api.ERROR_NO_IMAGES_FOUND and dec(buf=pj) == api.ERROR_GAIN_MAP_IMAGE_NOT_FOUND
    assert dec(buf=np.concatenate([pj, gj])) == api.ERROR_METADATA_ERROR          # two JPEGs, no gain-map XMP
    assert dec(fmt=api.OUTPUT_SDR) == api.ERROR_INSUFFICIENT_RESOURCE and (dest.width, dest.height) == (1280, 720)   # SDR rendition: RGBA8888, same size query
    assert dec() == api.ERROR_INSUFFICIENT_RESOURCE and (dest.width, dest.height, dest.colorGamut) == (1280, 720, api.CG_BT709)   # the size query
    assert abs(dmd.maxContentBoost - 10.0) < 1e-4 and dmd.version == b"1.0"


@pytest.mark.gpu
@pytest.mark.parametrize("device", [True, False])
def test_gpu_decode_batch_equals_single_decodes(hip, orc, device):
    """uhdr_hip_jpegr_decode_batch: files of different sizes, gamuts and qualities (plus a truncated one, an SDR-only JPEG and a size
    query in the middle) decoded in one call == the CPU restatement per file; the bad ones fail alone"""
    from oracle import jpegr_oracle as J
    from tests.gpu_util import dev_empty, stream_ptr, to_host
    from tests.test_gpu_parity import smooth_frame
    lib = hip.load()
    files, want = [], []
    for (w, h), sg, tf, q in (((640, 480), 0, 1, 95), ((200, 120), 1, 2, 80), ((72, 40), 2, 0, 100), ((1280, 720), 0, 1, 90), ((64, 64), 0, 2, 50)):
        p010, yuv = smooth_frame(w, h, w + q)
        files.append(J.encode_api1(p010, yuv, w, h, sg, hip.CG_BT2100, tf, q))
    files.append(open(SAMPLE, "rb").read())
    good = len(files)
    files.insert(2, files[0][:len(files[0]) * 3 // 5])          # truncated: the gain map is gone
    files.insert(4, J.find_images(files[0]) and files[0][:J.find_images(files[0])[0][1]])   # the primary image alone
    n = len(files)
    bufs = [np.frombuffer(f, np.uint8) for f in files]
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
    sizes = (C.c_size_t * n)(*[b.size for b in bufs])
    dests = (hip.Image * n)()
    mds = (hip.Metadata * n)()
    status = (C.c_int * n)()
    # 1st call: sizes only
    rc = lib.uhdr_hip_jpegr_decode_batch(n, ptrs, sizes, hip.OUTPUT_HDR_PQ, FLT_MAX, None, None, dests, mds, status, hip.APPLY_EXACT,
                                         hip.MEM_DEVICE if device else hip.MEM_HOST, None)
    assert rc != 0
    st = list(status)
    assert st[2] in (hip.ERROR_GAIN_MAP_IMAGE_NOT_FOUND, hip.ERROR_NO_IMAGES_FOUND, hip.ERROR_DECODE_ERROR) and st[4] == hip.ERROR_GAIN_MAP_IMAGE_NOT_FOUND
    assert all(s == hip.ERROR_INSUFFICIENT_RESOURCE for i, s in enumerate(st) if i not in (2, 4))
    need = [dests[i].width * dests[i].height * 4 if st[i] == hip.ERROR_INSUFFICIENT_RESOURCE else 0 for i in range(n)]
    need[5] = 0                                                 # ... and leave one of the good files as a size query
    outs = [(dev_empty(m, 0xCD) if device else np.full(max(m, 1), 0xCD, np.uint8)) if m else None for m in need]
    optr = (C.c_void_p * n)(*[(o.data_ptr() if device else o.ctypes.data) if o is not None else None for o in outs])
    ocap = (C.c_size_t * n)(*need)
    rc = lib.uhdr_hip_jpegr_decode_batch(n, ptrs, sizes, hip.OUTPUT_HDR_PQ, FLT_MAX, optr, ocap, dests, mds, status, hip.APPLY_EXACT,
                                         hip.MEM_DEVICE if device else hip.MEM_HOST, stream_ptr() if device else None)
    st = list(status)
    assert rc == st[2] != 0 and st[4] == hip.ERROR_GAIN_MAP_IMAGE_NOT_FOUND and st[5] == hip.ERROR_INSUFFICIENT_RESOURCE
    for i in range(n):
        if i in (2, 4, 5):
            continue
        assert st[i] == 0, (i, st[i])
        ost, ref, ow, oh, gamut, md = J.decode(files[i], orc.OUT_HDR_PQ, FLT_MAX)
        got = to_host(outs[i], need[i]) if device else outs[i][:need[i]]
        assert ost == 0 and (dests[i].width, dests[i].height, dests[i].colorGamut) == (ow, oh, gamut) and np.array_equal(got, ref), i
        assert mds[i].maxContentBoost == np.float32(md["max"])
    assert good == 6


@pytest.mark.gpu
def test_gpu_decodes_a_file_whose_primary_image_has_restart_intervals(hip, orc, tmp_path):
    """API-4 around an SDR JPEG written by libjpeg-turbo (Pillow) with DRI / RSTn markers and optimised tables -- what a camera or an
    editor hands over -- and API-3 deriving the gain map from it: both decode to the restatement's rendition"""
    import io
    try:
        from PIL import Image
    except ImportError:
        pytest.skip("no Pillow here")
    from oracle import jpegr_oracle as J
    from tests.test_gpu_parity import smooth_frame
    lib = hip.load()
    w, h = 640, 480
    p010, yuv = smooth_frame(w, h, 31)
    Y = yuv[:w * h].reshape(h, w)
    U = yuv[w * h:w * h * 5 // 4].reshape(h // 2, w // 2)
    V = yuv[w * h * 5 // 4:].reshape(h // 2, w // 2)
    ycc = np.stack([Y, np.repeat(np.repeat(U, 2, 0), 2, 1), np.repeat(np.repeat(V, 2, 0), 2, 1)], -1)
    b = io.BytesIO()
    Image.fromarray(ycc, mode="YCbCr").save(b, "JPEG", quality=92, subsampling=2, restart_marker_rows=1, optimize=True)
    sdr_jpeg = b.getvalue()
    assert b"\xff\xdd" in sdr_jpeg[:1500] and b"\xff\xd0" in sdr_jpeg
    want = J.encode_api3(p010, w, h, hip.CG_BT2100, sdr_jpeg, hip.CG_BT709, hip.TF_HLG)
    assert isinstance(want, bytes)
    e = _Enc(hip, True)
    rc, got = e.run("api3", e.p010(p010, w, h, hip.CG_BT2100), sdr_jpeg, hip.CG_BT709, hip.TF_HLG)
    assert rc == 0 and got == want
    st, ref, ow, oh, gamut, md = J.decode(got, orc.OUT_HDR_HLG, FLT_MAX)
    for device in (True, False):
        rc, dec, dest, _ = _gpu_decode(lib, hip, got, hip.OUTPUT_HDR_HLG, FLT_MAX, hip.APPLY_EXACT, device)
        assert rc == st == 0 and (dest.width, dest.height, dest.colorGamut) == (w, h, hip.CG_BT709) and np.array_equal(dec, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("device", [True, False])
def test_gpu_sdr_rendition_is_libjpeg_turbos(hip, orc, device):
    """decodeJPEGR(ULTRAHDR_OUTPUT_SDR): the primary image as RGBA8888 == the restatement of libjpeg-turbo's DECODE_TO_RGBA path, and --
    where Pillow (which bundles libjpeg-turbo) is usable -- == what libjpeg-turbo itself makes of the reference's own sample file"""
    from oracle import jpegr_oracle as J
    from tests.test_gpu_parity import smooth_frame
    lib = hip.load()
    data = open(SAMPLE, "rb").read()
    files = [data]
    for (w, h), sg, tf, q in ENC_CASES + (((8, 8), 0, 1, 90), ((1280, 720), 1, 1, 97)):
        p010, yuv = smooth_frame(w, h, w + 3)
        files.append(J.encode_api1(p010, yuv, w, h, sg, hip.CG_BT2100, tf, q))
    for k, blob in enumerate(files):
        st, want, w, h, gamut, _ = J.decode(blob, orc.OUT_SDR, FLT_MAX)
        rc, got, dest, _ = _gpu_decode(lib, hip, blob, hip.OUTPUT_SDR, FLT_MAX, hip.APPLY_FAST, device)
        assert rc == st == 0 and (dest.width, dest.height, dest.colorGamut) == (w, h, gamut), k
        assert np.array_equal(got, want), (k, int((got != want).sum()))
        assert np.all(got.reshape(h, w, 4)[..., 3] == 0xFF)
    try:
        import io
        from PIL import Image
    except ImportError:
        return
    rc, got, dest, _ = _gpu_decode(lib, hip, data, hip.OUTPUT_SDR, FLT_MAX, hip.APPLY_FAST, device)
    turbo = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    assert np.array_equal(got.reshape(720, 1280, 4)[..., :3], turbo)


@pytest.mark.gpu
def test_gpu_codec_random_sweep(hip, orc):
    """the reference's fuzzers draw API / gamuts / transfer function / quality / dimensions at random (fuzzer/ultrahdr_enc_fuzzer.cpp:87-319);
    the same sweep as a parity test: 160 random configurations, every file byte for byte against the CPU restatement, then decoded to a
    random output format and compared again"""
    from oracle import jpegr_oracle as J
    from tests.test_gpu_parity import smooth_frame
    lib = hip.load()
    rng = np.random.RandomState(20261004)
    fmts = (hip.OUTPUT_SDR, hip.OUTPUT_HDR_LINEAR, hip.OUTPUT_HDR_PQ, hip.OUTPUT_HDR_HLG)
    ofmt = {hip.OUTPUT_SDR: orc.OUT_SDR, hip.OUTPUT_HDR_LINEAR: orc.OUT_HDR_LINEAR, hip.OUTPUT_HDR_PQ: orc.OUT_HDR_PQ, hip.OUTPUT_HDR_HLG: orc.OUT_HDR_HLG}
    for case in range(160):
        w, h = 2 * int(rng.randint(4, 200)), 2 * int(rng.randint(4, 150))
        if case % 6 == 0:
            w, h = 4 * int(rng.randint(2, 100)), 4 * int(rng.randint(2, 75))
        pg, sg, tf, q = int(rng.randint(0, 3)), int(rng.randint(0, 3)), int(rng.randint(0, 3)), int(rng.randint(0, 101))
        api_n, device = int(rng.randint(0, 4)), bool(rng.randint(0, 2))
        p010, yuv = smooth_frame(w, h, 1000 + case)
        if case % 5 == 0:       # noise instead of smooth content
            yuv = rng.randint(0, 256, yuv.size).astype(np.uint8)
            p010 = (rng.randint(64, 941, p010.size).astype(np.uint16) << 6)
        e = _Enc(hip, device)
        if api_n == 0:
            want = J.encode_api0(p010, w, h, pg, tf, q)
            rc, got = e.run("api0", e.p010(p010, w, h, pg), tf, q, None)
        elif api_n == 1:
            want = J.encode_api1(p010, yuv, w, h, sg, pg, tf, q)
            rc, got = e.run("api1", e.p010(p010, w, h, pg), e.yuv(yuv, w, h, sg), tf, q, None)
        else:
            sdr_jpeg = orc.jpeg_encode("orc", yuv[:w * h], yuv[w * h:], w, h, q, icc=J.icc_profile_srgb_transfer(sg) if case % 2 else None)
            if api_n == 2:
                want = J.encode_api2(p010, yuv, w, h, sg, pg, sdr_jpeg, sg, tf)
                rc, got = e.run("api2", e.p010(p010, w, h, pg), e.yuv(yuv, w, h, sg), sdr_jpeg, sg, tf)
            else:
                want = J.encode_api3(p010, w, h, pg, sdr_jpeg, sg, tf)
                rc, got = e.run("api3", e.p010(p010, w, h, pg), sdr_jpeg, sg, tf)
        tag = (case, api_n, w, h, pg, sg, tf, q, device)
        assert isinstance(want, bytes) and rc == 0 and got == want, tag
        fmt = fmts[int(rng.randint(0, 4))]
        boost = float(rng.choice([FLT_MAX, 1.0, 2.5]))
        st, ref, ow, oh, gamut, md = J.decode(got, ofmt[fmt], boost)
        rc, dec, dest, _ = _gpu_decode(lib, hip, got, fmt, boost, hip.APPLY_EXACT, device)
        assert rc == st, tag + (fmt, boost, rc, st)
        if st == 0:
            assert (dest.width, dest.height) == (w, h) and np.array_equal(dec, ref), tag + (fmt, boost)
        else:   # a width or height that is not a multiple of 4 encodes, but its map no longer divides the image (ultrahdr.cpp:388-406)
            assert st == hip.ERROR_UNSUPPORTED_MAP_SCALE_FACTOR and (w % 4 or h % 4) and fmt != hip.OUTPUT_SDR, tag


@pytest.mark.gpu
def test_gpu_entry_points_are_safe_to_call_from_several_threads(hip, orc):
    """four host threads encode, decode and run generate / apply on their own images at once (ctypes releases the GIL): the shared
    staging buffers and workspaces are serialised inside the library, every result must still be the single-threaded one"""
    import threading
    from oracle import jpegr_oracle as J
    from tests.test_gpu_parity import smooth_frame
    lib = hip.load()
    jobs = []
    for t, ((w, h), sg, tf, q) in enumerate((((640, 480), 0, 1, 95), ((200, 120), 1, 2, 80), ((320, 240), 2, 0, 90), ((1280, 720), 0, 1, 85))):
        p010, yuv = smooth_frame(w, h, 50 + t)
        want = J.encode_api1(p010, yuv, w, h, sg, hip.CG_BT2100, tf, q)
        st, ref, _, _, _, _ = J.decode(want, orc.OUT_HDR_HLG, FLT_MAX)
        assert st == 0
        jobs.append((w, h, sg, tf, q, p010, yuv, want, ref))
    errors = []

    def worker(k):
        try:
            w, h, sg, tf, q, p010, yuv, want, ref = jobs[k]
            out, n = np.zeros(w * h * 3 + 65536, np.uint8), C.c_size_t()
            pi, yi = hip.p010_image(p010.ctypes.data, w, h, hip.CG_BT2100), hip.yuv420_image(yuv.ctypes.data, w, h, sg)
            wb = np.frombuffer(want, np.uint8)
            rend = np.zeros(w * h * 4, np.uint8)
            for it in range(12):
                rc = lib.uhdr_hip_jpegr_encode_api1(C.byref(pi), C.byref(yi), tf, q, None, 0, C.c_void_p(out.ctypes.data), out.size, C.byref(n), hip.MEM_HOST, None)
                if rc != 0 or out[:n.value].tobytes() != want:
                    errors.append((k, it, "encode", rc))
                    return
                dest, md = hip.Image(), hip.Metadata()
                rc = lib.uhdr_hip_jpegr_decode(C.c_void_p(wb.ctypes.data), wb.size, hip.OUTPUT_HDR_HLG, FLT_MAX, C.c_void_p(rend.ctypes.data), rend.size, C.byref(dest),
                                               C.byref(md), hip.APPLY_EXACT, hip.MEM_HOST, None)
                if rc != 0 or not np.array_equal(rend, ref):
                    errors.append((k, it, "decode", rc))
                    return
        except Exception as e:   # noqa: BLE001
            errors.append((k, repr(e)))
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors and not any(t.is_alive() for t in threads), errors
