"""CPU restatement of JpegR::decodeJPEGR (lib/src/jpegr.cpp:655-822) for the HDR output formats -- TEST INFRASTRUCTURE ONLY.

Container scan (extractPrimaryImageAndGainMap, jpegr.cpp:823-876), XMP metadata (getMetadataFromXMP, jpegrutils.cpp:436-545
with the XMPXmlHandler getters :213-330) and the ICC gamut (IccHelper::readIccColorGamut, icc.cpp:615-685) in Python (a few
hundred bytes of bookkeeping); the pixels go through the C checkers (orc_jpeg_decode, orc_applyGainMap), each pinned on its
own.  Parity status of the CONTAINER level: unpinned -- the reference's decodeJPEGR cannot be built here (jpegr.cpp needs
libjpeg-turbo headers and the un-vendored libheif fork) and its tests hold no decoded output for tests/data/sample_jpegr.jpeg;
what is pinned is every stage below it and the file's own metadata (hdrgm:GainMapMax = 3.32193 -> 10.0)."""
import math
import re
import struct

import numpy as np

from . import oracle as O

XMP_NS = b"http://ns.adobe.com/xap/1.0/\0"
ICC_ID = b"ICC_PROFILE\0"


def _walk(d, pos):
    """index after the EOI of the image whose SOI sits at pos, or 0"""
    n = len(d)
    if d[pos:pos + 2] != b"\xff\xd8":
        return 0
    pos += 2
    while True:
        while pos + 1 < n and d[pos] == 0xFF and d[pos + 1] == 0xFF:
            pos += 1
        if pos + 2 > n or d[pos] != 0xFF:
            return 0
        m = d[pos + 1]
        if m == 0xD9:
            return pos + 2
        if m == 0x01 or 0xD0 <= m <= 0xD7:
            pos += 2
            continue
        if pos + 4 > n:
            return 0
        ln = (d[pos + 2] << 8) | d[pos + 3]
        if ln < 2 or pos + 2 + ln > n:
            return 0
        pos += 2 + ln
        if m == 0xDA:
            while True:
                pos = d.find(b"\xff", pos, n - 1)
                if pos < 0:
                    return 0
                k = d[pos + 1]
                if k == 0 or 0xD0 <= k <= 0xD7:
                    pos += 2
                elif k == 0xFF:
                    pos += 1
                else:
                    break


def find_images(d):
    out, pos = [], 0
    while len(out) < 2:
        pos = d.find(b"\xff\xd8", pos)
        if pos < 0:
            break
        end = _walk(d, pos)
        if end == 0:
            break
        out.append((pos, end - pos))
        pos = end
    return out


def app_segment(jpg, marker, prefix):
    pos = 2
    while pos + 4 <= len(jpg) and jpg[pos] == 0xFF:
        m = jpg[pos + 1]
        if m in (0xDA, 0xD9):
            break
        ln = (jpg[pos + 2] << 8) | jpg[pos + 3]
        if m == marker and ln - 2 > len(prefix) and jpg[pos + 4:pos + 4 + len(prefix)] == prefix:
            return jpg[pos + 4:pos + 2 + ln]
        pos += 2 + ln
    return None


def _attr(xml, name):
    m = re.search(r"(?<![\w:\-])" + re.escape(name) + r"\s*=\s*(?:\"([^\"]*)\"|'([^']*)')", xml)
    return None if m is None else (m.group(1) if m.group(1) is not None else m.group(2))


def _float(s):
    m = re.match(r"\s*[+-]?(?:\d+\.?\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?|inf|nan)", s)
    return None if m is None else np.float32(float(m.group(0)))


def metadata_from_xmp(payload):
    if payload is None or not payload.startswith(XMP_NS[:-1]):
        return None
    xml = payload[len(XMP_NS):].decode("latin1")
    md = dict(version=_attr(xml, "hdrgm:Version"))
    if md["version"] is None:
        return None
    req = {k: _attr(xml, "hdrgm:" + k) for k in ("GainMapMax", "HDRCapacityMax")}
    if any(v is None or _float(v) is None for v in req.values()):
        return None
    ex = lambda v: np.float32(math.pow(2.0, float(v)))      # (float)exp2((double)val)
    md["max"], md["capmax"] = ex(_float(req["GainMapMax"])), ex(_float(req["HDRCapacityMax"]))
    for key, name, default, log in (("min", "GainMapMin", 1.0, True), ("gamma", "Gamma", 1.0, False), ("off_sdr", "OffsetSDR", 1 / 64, False),
                                    ("off_hdr", "OffsetHDR", 1 / 64, False), ("capmin", "HDRCapacityMin", 1.0, True)):
        v = _attr(xml, "hdrgm:" + name)
        if v is None:
            md[key] = np.float32(default)
        else:
            f = _float(v)
            if f is None:
                return None
            md[key] = ex(f) if log else f
    b = _attr(xml, "hdrgm:BaseRenditionIsHDR")
    if b is not None and b != "False":
        return None
    return md


def gamut_from_icc(payload):
    if payload is None or len(payload) < 132 + 14 or not payload.startswith(ICC_ID):
        return O.CG_UNSPECIFIED
    icc = payload[14:]
    tags, = struct.unpack(">I", icc[128:132])
    found = {}
    for t in range(tags):
        if len(icc) < 132 + (t + 1) * 12:
            return O.CG_UNSPECIFIED
        sig, off, ln = struct.unpack(">4sII", icc[132 + 12 * t:144 + 12 * t])
        if sig in (b"rXYZ", b"gXYZ", b"bXYZ") and sig not in found:
            found[sig] = (off, ln)
    cols = []
    for sig in (b"rXYZ", b"gXYZ", b"bXYZ"):
        if sig not in found or found[sig][0] == 0 or found[sig][1] != 20 or found[sig][0] + 20 > len(icc):
            return O.CG_UNSPECIFIED
        cols.append(icc[found[sig][0]:found[sig][0] + 20])
    fx = lambda x: int(math.floor(float(np.float32(x)) * 65536.0 + 0.5))
    ff = lambda v: np.float32(v) * np.float32(1.0 / 65536.0)
    mats = {O.CG_BT709: [[ff(0x6FA2), ff(0x6299), ff(0x24A0)], [ff(0x38F5), ff(0xB785), ff(0x0F84)], [ff(0x0390), ff(0x18DA), ff(0xB6CF)]],
            O.CG_P3: [[0.515102, 0.291965, 0.157153], [0.241182, 0.692236, 0.0665819], [-0.00104941, 0.0418818, 0.784378]],
            O.CG_BT2100: [[0.673459, 0.165661, 0.125100], [0.279033, 0.675338, 0.0456288], [-0.00193139, 0.0299794, 0.797162]]}
    for g, m in mats.items():
        want = [b"XYZ \0\0\0\0" + struct.pack(">iii", fx(m[0][c]), fx(m[1][c]), fx(m[2][c])) for c in range(3)]
        if cols == want:
            return g
    return O.CG_UNSPECIFIED


def decode(data, output_format, max_display_boost, threads=8):
    """-> (status, out bytes ndarray, w, h, gamut, metadata dict); status values are the reference's"""
    if max_display_boost < 1.0:
        return -10008, None, 0, 0, -1, None
    imgs = find_images(data)
    if not imgs:
        return -20006, None, 0, 0, -1, None
    if len(imgs) == 1:
        return -20003, None, 0, 0, -1, None
    pj, gj = data[imgs[0][0]:imgs[0][0] + imgs[0][1]], data[imgs[1][0]:imgs[1][0] + imgs[1][1]]
    st, planes, w, h, gray = O.jpeg_decode("orc", pj)
    if st <= 0 or gray:
        return -20002, None, 0, 0, -1, None
    gst, gplanes, gw, gh, ggray = O.jpeg_decode("orc", gj)
    if gst <= 0:
        return -20002, None, 0, 0, -1, None
    md = metadata_from_xmp(app_segment(gj, 0xE1, XMP_NS))
    if md is None:
        return -20005, None, w, h, -1, None
    gamut = gamut_from_icc(app_segment(pj, 0xE2, ICC_ID))
    omd = O.Metadata(float(md["max"]), float(md["min"]), float(md["gamma"]), float(md["off_sdr"]), float(md["off_hdr"]), float(md["capmin"]),
                     float(md["capmax"]), 1 if md["version"] == "1.0" else 0)
    yi = O.yuv420_image(planes, w, h, gamut)
    gmap = np.ascontiguousarray(gplanes[:gw * gh].reshape(gh, gw))
    ast, out, dest = O.apply("orc_", yi, gmap, omd, output_format, max_display_boost, threads=threads)
    return ast, out, w, h, gamut, md
