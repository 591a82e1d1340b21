"""CPU restatement of JpegR::decodeJPEGR (lib/src/jpegr.cpp:655-822) for the HDR output formats -- TEST INFRASTRUCTURE ONLY.

Container scan (extractPrimaryImageAndGainMap, jpegr.cpp:823-876), XMP metadata (getMetadataFromXMP, jpegrutils.cpp:436-545
with the XMPXmlHandler getters :213-330) and the ICC gamut (IccHelper::readIccColorGamut, icc.cpp:615-685) in Python (a few
hundred bytes of bookkeeping); the pixels go through the C checkers (orc_jpeg_decode, orc_applyGainMap), each pinned on its
own.  Parity status of the CONTAINER level: unpinned -- the reference's decodeJPEGR cannot be built here (jpegr.cpp needs
libjpeg-turbo headers and the un-vendored libheif fork) and its tests hold no decoded output for tests/data/sample_jpegr.jpeg;
what is pinned is every stage below it and the file's own metadata (hdrgm:GainMapMax = 3.32193 -> 10.0)."""
import ctypes
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


# glibc's log2f / exp2f: what the reference's `log2(metadata.maxContentBoost)` / `exp2(val)` on float operands bind to (pinned
# against its object code, tests/test_ref_container.py)
_libm = ctypes.CDLL("libm.so.6")
_libm.log2f.restype = _libm.exp2f.restype = ctypes.c_float
_libm.log2f.argtypes = _libm.exp2f.argtypes = [ctypes.c_float]


def _log2f(v):
    return float(np.float32(_libm.log2f(float(np.float32(v)))))


def _exp2f(v):
    return np.float32(_libm.exp2f(float(np.float32(v))))


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
    ex = _exp2f                                               # exp2(float val): the float overload (jpegrutils.cpp:225-229)
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
    if st == -2 and O.load_libjpeg() is not None:   # a progressive primary image: outside the baseline restatement, libjpeg itself decodes it
        st, planes, w, h, gray = O.jpeg_decode("lj", pj)
    if st <= 0 or gray:
        return -20002, None, 0, 0, -1, None
    if output_format == O.OUT_SDR:      # jpegr.cpp:768-786: the primary image through libjpeg-turbo's DECODE_TO_RGBA, nothing else
        if (w | h) & 1:
            return -30000, None, w, h, -1, None
        return 0, O.ycc420_to_rgba(planes, w, h).reshape(-1), w, h, gamut_from_icc(app_segment(pj, 0xE2, ICC_ID)), None
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


# ---------------------------------------------------------------------------------------------------------------------
# assembly side: generateXmpFor{Primary,Secondary}Image (jpegrutils.cpp:547-611), generateMpf (multipictureformat.cpp:30-92),
# IccHelper::writeIccProfile for the sRGB transfer (icc.cpp:410-600), appendGainMap (jpegr.cpp:951-1130) and encodeJPEGR
# API-1 (jpegr.cpp:249-381).  Pinned: re-assembling the two JPEG streams found in the reference's own sample file gives back
# the sample file, byte for byte, and the sample's ICC segment is the sRGB/sRGB profile (tests/test_jpegr_container.py).
# ---------------------------------------------------------------------------------------------------------------------
_HEAD = ('<x:xmpmeta\n  xmlns:x="adobe:ns:meta/"\n  x:xmptk="Adobe XMP Core 5.1.2">\n  <rdf:RDF\n'
         '    xmlns:rdf="http://www.w3.org/1999/02/22-rdf-syntax-ns#">\n    <rdf:Description\n')


def _g(v):
    return "%g" % float(v)


def xmp_primary(secondary_length, version="1.0"):
    item = ('          <rdf:li\n            rdf:parseType="Resource">\n            <Container:Item\n              Item:Semantic="%s"\n'
            '              Item:Mime="image/jpeg"%s/>\n          </rdf:li>\n')
    return (_HEAD + '      xmlns:Container="http://ns.google.com/photos/1.0/container/"\n      xmlns:Item="http://ns.google.com/photos/1.0/container/item/"\n'
            '      xmlns:hdrgm="http://ns.adobe.com/hdr-gain-map/1.0/"\n      hdrgm:Version="%s">\n      <Container:Directory>\n        <rdf:Seq>\n' % version
            + item % ("Primary", "") + item % ("GainMap", '\n              Item:Length="%d"' % secondary_length)
            + '        </rdf:Seq>\n      </Container:Directory>\n    </rdf:Description>\n  </rdf:RDF>\n</x:xmpmeta>\n')


def xmp_secondary(md):
    l2 = lambda v: _g(_log2f(v))                              # log2(float): the float overload (jpegrutils.cpp:598-604)
    rows = [("Version", md["version"]), ("GainMapMin", l2(md["min"])), ("GainMapMax", l2(md["max"])), ("Gamma", _g(np.float32(md["gamma"]))),
            ("OffsetSDR", _g(np.float32(md["off_sdr"]))), ("OffsetHDR", _g(np.float32(md["off_hdr"]))), ("HDRCapacityMin", l2(md["capmin"])),
            ("HDRCapacityMax", l2(md["capmax"])), ("BaseRenditionIsHDR", "False")]
    return (_HEAD + '      xmlns:hdrgm="http://ns.adobe.com/hdr-gain-map/1.0/"' + "".join('\n      hdrgm:%s="%s"' % r for r in rows)
            + "/>\n  </rdf:RDF>\n</x:xmpmeta>\n")


def mpf(primary_size, primary_offset, secondary_size, secondary_offset):
    b = b"MPF\0MM\0*" + struct.pack(">IH", 8, 3)
    b += struct.pack(">HHI4s", 0xB000, 7, 4, b"0100") + struct.pack(">HHII", 0xB001, 4, 1, 2) + struct.pack(">HHI", 0xB002, 7, 32)
    b += struct.pack(">I", len(b) - 4 + 8) + struct.pack(">I", 0)
    b += struct.pack(">IIIHH", 0x030000, primary_size, primary_offset, 0, 0) + struct.pack(">IIIHH", 0, secondary_size, secondary_offset, 0, 0)
    return b


def icc_profile_srgb_transfer(gamut):
    fx = lambda x: int(math.floor(float(np.float32(x)) * 65536.0 + 0.5)) & 0xFFFFFFFF
    ff = lambda v: np.float32(v) * np.float32(1.52587890625e-5)
    mats = {O.CG_BT709: ("sRGB", [[ff(0x6FA2), ff(0x6299), ff(0x24A0)], [ff(0x38F5), ff(0xB785), ff(0x0F84)], [ff(0x0390), ff(0x18DA), ff(0xB6CF)]]),
            O.CG_P3: ("Display P3", [[0.515102, 0.291965, 0.157153], [0.241182, 0.692236, 0.0665819], [-0.00104941, 0.0418818, 0.784378]]),
            O.CG_BT2100: ("Rec2020", [[0.673459, 0.165661, 0.125100], [0.279033, 0.675338, 0.0456288], [-0.00193139, 0.0299794, 0.797162]])}
    name, m = mats[gamut]

    def text(t):
        b = b"mluc" + struct.pack(">IIIII", 0, 1, 12, 0x656E5553, 2 * len(t)) + struct.pack(">I", 28) + t.encode("utf-16-be")
        return b.ljust(((2 * len(t) + 28 + 2) >> 2) << 2, b"\0")

    xyz = lambda x, y, z: b"XYZ " + struct.pack(">IIII", 0, fx(x), fx(y), fx(z))
    para = b"para" + struct.pack(">IHH", 0, 4, 0) + b"".join(struct.pack(">I", fx(v)) for v in
                                                            (2.4, np.float32(1 / 1.055), np.float32(0.055 / 1.055), np.float32(1 / 12.92), 0.04045, 0.0, 0.0))
    tags = [(b"desc", text(name + " Gamut with sRGB Transfer")), (b"rXYZ", xyz(m[0][0], m[1][0], m[2][0])), (b"gXYZ", xyz(m[0][1], m[1][1], m[2][1])),
            (b"bXYZ", xyz(m[0][2], m[1][2], m[2][2])), (b"wtpt", xyz(0.9642, 1.0, 0.8249)), (b"rTRC", para), (b"gTRC", para), (b"bTRC", para),
            (b"cprt", text("Google Inc. 2022"))]
    table = 12 * len(tags)
    size = 132 + table + sum(len(t[1]) for t in tags)
    hdr = struct.pack(">III4s4s4s", size, 0, 0x04300000, b"mntr", b"RGB ", b"XYZ ") + b"\0" * 12 + b"acsp" + b"\0" * 24
    hdr += struct.pack(">IIII", 1, fx(0.9642), fx(1.0), fx(0.8249)) + b"\0" * 48 + struct.pack(">I", len(tags))
    out, off = b"ICC_PROFILE\0\x01\x01" + hdr, 132 + table
    for sig, body in tags:
        out += sig + struct.pack(">II", off, len(body))
        off += len(body)
    return out + b"".join(t[1] for t in tags)


def _header_segments(d):
    """jpeg_read_header's walk: [(marker, data offset, data length)] up to SOS, or None where it would fail"""
    if len(d) < 4 or d[0] != 0xFF or d[1] != 0xD8:
        return None
    p, out, sof = 2, [], False
    while True:
        while p < len(d) and d[p] != 0xFF:
            p += 1
        while p < len(d) and d[p] == 0xFF:
            p += 1
        if p >= len(d):
            return None
        m = d[p]
        p += 1
        if m in (0x00, 0x01) or 0xD0 <= m <= 0xD7:
            continue
        if m == 0xD9 or p + 2 > len(d):
            return None
        ln = (d[p] << 8) | d[p + 1]
        if ln < 2 or p + ln > len(d):
            return None
        if m == 0xDA:
            return out if sof else None
        if m in (0xC0, 0xC1, 0xC2):
            sof = True
        out.append((m, p + 2, ln - 2))
        p += ln


def extract_exif(d):
    """JpegDecoderHelper::extractEXIF (jpegdecoderhelper.cpp:146-188): (ok, exif_pos, exif bytes); the position counts the saved
    (APP0 / APP1) segments only"""
    segs = _header_segments(d)
    if segs is None:
        return False, -1, None
    pos = 2
    for m, off, ln in segs:
        if m not in (0xE0, 0xE1):
            continue
        pos += 4 + ln
        if m == 0xE1 and ln > 6 and d[off:off + 6] == b"Exif\0\0":
            return True, pos - ln, d[off:off + ln]
    return True, -1, None


def append_gainmap(primary, gainmap, md, exif=None, icc=None):
    """-> bytes, or a negative status"""
    if md["version"] != "1.0" or md["max"] < md["min"] or md["capmax"] < md["capmin"] or md["capmin"] < 1.0 or md["off_sdr"] < 0 or md["off_hdr"] < 0 \
            or md["gamma"] <= 0:
        return -10010
    ns = XMP_NS
    xs = xmp_secondary(md).encode()
    xs_len = 2 + len(ns) + len(xs)
    secondary_size = 2 + xs_len + len(gainmap)
    xp = xmp_primary(secondary_size, md["version"]).encode()
    ok, epos, inside = extract_exif(primary)
    if not ok:
        return -20002
    if epos >= 0:
        if exif is not None:
            return -20007
        primary = primary[:epos - 4] + primary[epos + len(inside):]          # copyJpegWithoutExif, jpegr.cpp:63-73
        exif = inside
    out = b"\xff\xd8"
    if exif is not None:
        out += b"\xff\xe1" + struct.pack(">H", 2 + len(exif)) + exif
    out += b"\xff\xe1" + struct.pack(">H", 2 + len(ns) + len(xp)) + ns + xp
    if icc:
        out += b"\xff\xe2" + struct.pack(">H", 2 + len(icc)) + icc
    pos, length = len(out), 2 + 86
    primary_size = pos + length + len(primary)
    out += b"\xff\xe2" + struct.pack(">H", length) + mpf(primary_size, 0, secondary_size, primary_size - pos - 8)
    out += primary[2:] + b"\xff\xd8\xff\xe1" + struct.pack(">H", xs_len) + ns + xs + gainmap[2:]
    return out


def _md(omd):
    return dict(version="1.0", max=omd.maxContentBoost, min=omd.minContentBoost, gamma=omd.gamma, off_sdr=omd.offsetSdr, off_hdr=omd.offsetHdr,
                capmin=omd.hdrCapacityMin, capmax=omd.hdrCapacityMax)


def _gainmap_jpeg(p010, yuv_img, w, h, hdr_gamut, hdr_tf, sdr_is_601, threads):
    st, gmap, omd = O.generate("orc_", yuv_img, O.p010_image(p010, w, h, hdr_gamut), hdr_tf, sdr_is_601=sdr_is_601, threads=threads)
    assert st == 0
    return O.jpeg_encode("orc", np.ascontiguousarray(gmap.reshape(-1)), None, w // 4, h // 4, 85), _md(omd)


def _finish_from_planes(enc, ls, w, h, sdr_gamut, quality, gm_jpeg, md, exif):
    cs = ls >> 1
    if sdr_gamut != O.CG_P3:
        import ctypes as C
        img = O.yuv420_image(enc, w, h, sdr_gamut, ls, cs)
        assert O.load().orc_convertYuv(C.byref(img), sdr_gamut, O.CG_P3) == 0
    sdr_jpeg = O.jpeg_encode("orc", enc[:ls * h], enc[ls * h:], w, h, quality, ls, cs, icc=icc_profile_srgb_transfer(sdr_gamut))
    return append_gainmap(sdr_jpeg, gm_jpeg, md, exif=exif)


def encode_api0(p010, w, h, hdr_gamut, hdr_tf, quality, exif=None, threads=8):
    """encodeJPEGR API-0 (jpegr.cpp:186-247), tightly packed P010"""
    import ctypes as C
    ls = (w + 15) // 16 * 16
    enc = np.zeros(ls * h * 3 // 2, np.uint8)
    src, dst = O.p010_image(p010, w, h, hdr_gamut), O.yuv420_image(enc, w, h, hdr_gamut, ls, ls >> 1)
    assert O.load().orc_toneMap(C.byref(src), C.byref(dst)) == 0
    gm_jpeg, md = _gainmap_jpeg(p010, dst, w, h, hdr_gamut, hdr_tf, False, threads)
    return _finish_from_planes(enc, ls, w, h, hdr_gamut, quality, gm_jpeg, md, exif)


def encode_api1(p010, yuv, w, h, sdr_gamut, hdr_gamut, hdr_tf, quality, exif=None, threads=8):
    """encodeJPEGR API-1 (jpegr.cpp:249-381), tightly packed inputs: -> JPEG/R bytes"""
    gm_jpeg, md = _gainmap_jpeg(p010, O.yuv420_image(yuv, w, h, sdr_gamut), w, h, hdr_gamut, hdr_tf, False, threads)
    enc, ls = yuv.copy(), w
    if sdr_gamut != O.CG_P3:
        ls = (w + 15) // 16 * 16
        cs, cw, ch = ls >> 1, w // 2, h // 2
        enc = np.zeros(ls * h * 3 // 2, np.uint8)
        enc[:ls * h].reshape(h, ls)[:, :w] = yuv[:w * h].reshape(h, w)
        enc[ls * h:ls * h + cs * ch].reshape(ch, cs)[:, :cw] = yuv[w * h:w * h + cw * ch].reshape(ch, cw)
        enc[ls * h + cs * ch:].reshape(ch, cs)[:, :cw] = yuv[w * h + cw * ch:].reshape(ch, cw)
    return _finish_from_planes(enc, ls, w, h, sdr_gamut, quality, gm_jpeg, md, exif)


def encode_api4(sdr_jpeg, sdr_jpeg_gamut, gm_jpeg, md):
    """encodeJPEGR API-4 (jpegr.cpp:502-560)"""
    if _header_segments(sdr_jpeg) is None:
        return -20002
    icc = None
    if app_segment(sdr_jpeg, 0xE2, ICC_ID) is None:
        if not 0 <= sdr_jpeg_gamut <= 2:
            return -10003
        icc = icc_profile_srgb_transfer(sdr_jpeg_gamut)
    return append_gainmap(sdr_jpeg, gm_jpeg, md, icc=icc)


def encode_api2(p010, yuv, w, h, sdr_gamut, hdr_gamut, sdr_jpeg, sdr_jpeg_gamut, hdr_tf, threads=8):
    """encodeJPEGR API-2 (jpegr.cpp:384-437)"""
    gm_jpeg, md = _gainmap_jpeg(p010, O.yuv420_image(yuv, w, h, sdr_gamut), w, h, hdr_gamut, hdr_tf, False, threads)
    return encode_api4(sdr_jpeg, sdr_jpeg_gamut, gm_jpeg, md)


def encode_api3(p010, w, h, hdr_gamut, sdr_jpeg, sdr_jpeg_gamut, hdr_tf, threads=8):
    """encodeJPEGR API-3 (jpegr.cpp:439-500)"""
    st, planes, jw, jh, gray = O.jpeg_decode("orc", sdr_jpeg)
    if st <= 0 or gray:
        return -20002
    icc = app_segment(sdr_jpeg, 0xE2, ICC_ID)
    if icc is not None:
        cg = gamut_from_icc(icc)
        if cg == O.CG_UNSPECIFIED or (sdr_jpeg_gamut != O.CG_UNSPECIFIED and sdr_jpeg_gamut != cg):
            return -10003
    else:
        if not 0 <= sdr_jpeg_gamut <= 2:
            return -10003
        cg = sdr_jpeg_gamut
    if (jw, jh) != (w, h):
        return -10006
    gm_jpeg, md = _gainmap_jpeg(p010, O.yuv420_image(planes, w, h, cg), w, h, hdr_gamut, hdr_tf, True, threads)
    return encode_api4(sdr_jpeg, sdr_jpeg_gamut, gm_jpeg, md)


def encode_apix(yuv, w, h, sdr_gamut, gmap, md, quality, exif=None):
    """encodeJPEGR "API-x" (jpegr.cpp:562-631): no BT.601 re-encode on this path"""
    gm_jpeg = O.jpeg_encode("orc", np.ascontiguousarray(gmap.reshape(-1)), None, gmap.shape[1], gmap.shape[0], 85)
    sdr_jpeg = O.jpeg_encode("orc", yuv[:w * h], yuv[w * h:], w, h, quality, icc=icc_profile_srgb_transfer(sdr_gamut))
    return append_gainmap(sdr_jpeg, gm_jpeg, md, exif=exif)


def info(data):
    """getJPEGRInfo (jpegr.cpp:633-653): -> status or [dict per image: offset, size, width, height, icc, exif, xmp (offset, size) ranges]"""
    imgs = find_images(data)
    if not imgs:
        return -20006
    if len(imgs) == 1:
        return -20003
    out = []
    for begin, ln in imgs[:2]:
        j = data[begin:begin + ln]
        segs = _header_segments(j)
        if segs is None:
            return -20002
        sof = [(off, n) for m, off, n in segs if 0xC0 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC)]
        hgt, wid = struct.unpack(">HH", j[sof[0][0] + 1:sof[0][0] + 5])
        if wid > 8192 or hgt > 8192:
            return -20002
        e = dict(offset=begin, size=ln, width=wid, height=hgt, icc=(0, 0), exif=(0, 0), xmp=(0, 0))
        for m, off, n in segs:                                   # jpegdecoderhelper.cpp:221-249
            if m not in (0xE1, 0xE2):
                continue
            if e["xmp"][1] == 0 and n > len(XMP_NS) and j[off:off + len(XMP_NS)] == XMP_NS:
                e["xmp"] = (begin + off, n)
            elif e["exif"][1] == 0 and n > 6 and j[off:off + 6] == b"Exif\0\0":
                e["exif"] = (begin + off, n)
            elif e["icc"][1] == 0 and n > len(ICC_ID) and j[off:off + len(ICC_ID)] == ICC_ID:
                e["icc"] = (begin + off, n)
        out.append(e)
    return out
