"""decodeJPEGR (lib/src/jpegr.cpp:655-822): container scan + XMP metadata + ICC gamut + two JPEG decodes + applyGainMap.

CPU part: the Python restatement (oracle/jpegr_oracle.py) on the reference's own sample file (tests/data/sample_jpegr.jpeg,
committed as a fixture) -- properties the file itself fixes.  GPU part: uhdr_hip_jpegr_decode against that restatement, on the
sample and on JPEG/R files assembled here from the device encoder's output.  The container level has no reference output to pin
against (the reference's decodeJPEGR is not buildable here and its tests keep no decoded bytes): parity of this level is
*unpinned*; every stage underneath (JPEG decoding, applyGainMap) is pinned on its own."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLT_MAX = 3.4028234663852886e38
SAMPLE = os.path.join(ROOT, "tests", "golden", "sample_jpegr.jpeg")


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
    assert _gpu_decode(lib, hip, data, hip.OUTPUT_SDR, FLT_MAX, 0, True)[0] == hip.ERROR_UNSUPPORTED_FEATURE
    assert _gpu_decode(lib, hip, gj + pj, hip.OUTPUT_HDR_HLG, FLT_MAX, 0, True)[0] == -20002                # primary is not 4:2:0: DECODE_ERROR
