"""ctypes binding of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
`load()` returns the C99 restatement (oracle/liboracle.so, prefix ``orc_``); `load_ref()` returns
the reference's own gainmapmath.cpp behind ref_harness.cpp (oracle/_ref/libuhdr_ref.so, prefix
``ref_``) or None when it has not been built (it is only buildable where /root/reference exists;
the prebuilt .so travels to the GPU box).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# enum values (lib/include/ultrahdr/ultrahdr.h:36-120)
CG_UNSPECIFIED, CG_BT709, CG_P3, CG_BT2100 = -1, 0, 1, 2
TF_LINEAR, TF_HLG, TF_PQ, TF_SRGB = 0, 1, 2, 3
OUT_SDR, OUT_HDR_LINEAR, OUT_HDR_PQ, OUT_HDR_HLG, OUT_HDR_LINEAR_RGB_10BIT = 0, 1, 2, 3, 4
FMT_P010, FMT_YUV420, FMT_MONOCHROME = 0, 1, 2


class Image(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_size_t), ("height", C.c_size_t),
                ("colorGamut", C.c_int32), ("chroma_data", C.c_void_p),
                ("luma_stride", C.c_size_t), ("chroma_stride", C.c_size_t),
                ("pixelFormat", C.c_int32)]


class Metadata(C.Structure):
    _fields_ = [("maxContentBoost", C.c_float), ("minContentBoost", C.c_float),
                ("gamma", C.c_float), ("offsetSdr", C.c_float), ("offsetHdr", C.c_float),
                ("hdrCapacityMin", C.c_float), ("hdrCapacityMax", C.c_float),
                ("version_ok", C.c_int32)]


class Effect(C.Structure):
    """type 0 crop(a=left,b=right,c=top,d=bottom) 1 mirror(a=dir) 2 rotate(a=degrees) 3 resize(a=w,b=h)"""
    _fields_ = [("type", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("c", C.c_int32), ("d", C.c_int32)]


class Color(C.Structure):
    _fields_ = [("r", C.c_float), ("g", C.c_float), ("b", C.c_float)]

    def tup(self):
        return (self.r, self.g, self.b)


def _declare(lib, p):
    f32, u8, sz, i32 = C.c_float, C.c_uint8, C.c_size_t, C.c_int
    IP, MP = C.POINTER(Image), C.POINTER(Metadata)

    def d(name, res, *args):
        fn = getattr(lib, p + name)
        fn.restype = res
        fn.argtypes = list(args)

    for n in ("srgbInvOetf", "hlgOetf", "hlgInvOetf", "pqOetf", "pqInvOetf"):
        d(n, f32, f32)
    d("luminance", f32, i32, Color)
    d("yuvToRgb", Color, i32, Color)
    d("rgbToYuv", Color, i32, Color)
    d("gamutConv", Color, i32, i32, Color, C.POINTER(C.c_int))
    d("yuvToYuv", Color, i32, i32, Color)
    d("encodeGain", u8, f32, f32, f32, f32, f32, f32)
    d("encodeGain3", u8, f32, f32, f32, f32)
    d("applyGain3", Color, Color, f32, f32, f32)
    d("applyGain4", Color, Color, f32, f32, f32, f32)
    d("getYuv420Pixel", Color, IP, sz, sz)
    d("getP010Pixel", Color, IP, sz, sz)
    d("sampleYuv420", Color, IP, sz, sz, sz)
    d("sampleP010", Color, IP, sz, sz, sz)
    d("fillShepardsIDW", None, C.POINTER(f32), i32, i32, i32)
    d("sampleMapIdw", f32, IP, sz, sz, sz)
    d("sampleMapFloat", f32, IP, f32, sz, sz)
    d("colorToRgba1010102", C.c_uint32, Color)
    d("colorToRgbaF16", C.c_uint64, Color)
    d("floatToHalf", C.c_uint16, f32)
    d("transformYuv420", None, IP, sz, sz, i32, i32)
    d("generateGainMap", i32, IP, IP, i32, MP, C.c_void_p, i32, i32)
    d("applyGainMap", i32, IP, IP, MP, i32, f32, IP, i32)
    d("generateGainMapLUT", i32, IP, IP, i32, MP, C.c_void_p, i32, i32)
    d("applyGainMapLUT", i32, IP, IP, MP, i32, f32, IP, i32)
    for n in ("srgbInvOetfLUT", "hlgOetfLUT", "hlgInvOetfLUT", "pqOetfLUT", "pqInvOetfLUT"):
        d(n, f32, f32)
    d("gainLutBuild", None, f32, f32, i32, f32, C.POINTER(f32))
    d("convertYuv", i32, IP, i32, i32)
    d("crop", i32, IP, i32, i32, i32, i32, IP)
    d("mirror", i32, IP, i32, IP)
    d("rotate", i32, IP, i32, IP)
    d("resize", i32, IP, i32, i32, IP)
    d("add_effects", i32, IP, C.c_void_p, i32, IP)
    if p == "orc_":
        d("generateGainMapStats", i32, IP, IP, i32, MP, C.c_void_p, i32, i32, C.POINTER(f32))
        d("toneMap", i32, IP, IP)
        d("fill_lcg", None, C.c_void_p, C.c_void_p, sz, sz, C.c_uint32)
        d("checksum_u8", C.c_uint64, C.c_void_p, sz)
        d("checksum_u32", C.c_uint64, C.c_void_p, sz)
        d("eval_transfer", None, i32, C.c_void_p, C.c_void_p, sz, f32, f32)
        d("jpeg_encode", C.c_long, C.c_void_p, C.c_void_p, i32, i32, i32, i32, i32, C.c_void_p, C.c_uint, C.c_void_p, C.c_long)
        d("jpeg_header", C.c_long, i32, i32, i32, i32, C.c_void_p, C.c_uint, C.c_void_p, C.c_long)
        d("jpeg_block_count", C.c_long, i32, i32, i32)
        d("jpeg_coefficients", C.c_long, C.c_void_p, C.c_void_p, i32, i32, i32, i32, i32, C.c_void_p)
        d("jpeg_quant_table", None, i32, i32, C.c_void_p)
        d("jpeg_decode", C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int))
        d("lut_table", C.POINTER(f32), i32, C.POINTER(sz))
        d("gainLutFactor", f32, C.POINTER(f32), f32)
        d("applyGainLUT", Color, Color, f32, C.POINTER(f32))
    else:
        d("applyGainLUT", Color, Color, f32, f32, f32, f32)
        d("gainLutFactor", f32, f32, f32, f32, f32)
    return lib


def build(ref=True):
    """Compile the checker (and, where /root/reference exists, oracle/_ref)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if ref and os.path.isdir("/root/reference/lib/src"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


_cache = {}


def load():
    if "orc" not in _cache:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        _cache["orc"] = _declare(C.CDLL(path), "orc_")
    return _cache["orc"]


def load_ref():
    if "ref" not in _cache:
        path = os.path.join(_HERE, "_ref", "libuhdr_ref.so")
        if not os.path.exists(path) and os.path.isdir("/root/reference/lib/src"):
            build(ref=True)
        _cache["ref"] = _declare(C.CDLL(path), "ref_") if os.path.exists(path) else None
    return _cache["ref"]


def load_libjpeg():
    """the image's libjpeg behind the reference's JpegEncoderHelper call sequence (oracle/jpeg_libjpeg_harness.c),
    or None where it cannot be built (no libjpeg)"""
    if "lj" not in _cache:
        path = os.path.join(_HERE, "libjpeg_harness.so")
        if not os.path.exists(path):
            try:
                subprocess.check_call(["make", "-s", "-C", _HERE, "jpeg"], stderr=subprocess.DEVNULL)
            except (subprocess.CalledProcessError, OSError):
                pass
        lib = None
        if os.path.exists(path):
            try:
                lib = C.CDLL(path)
                lib.lj_jpeg_encode.restype = C.c_long
                lib.lj_jpeg_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                               C.c_uint, C.c_void_p, C.c_long]
                lib.lj_jpeg_decode.restype = C.c_long
                lib.lj_jpeg_decode.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                               C.POINTER(C.c_int)]
            except OSError:
                lib = None
        _cache["lj"] = lib
    return _cache["lj"]


def jpeg_encode(which, y, uv, w, h, quality, luma_stride=None, chroma_stride=None, icc=None):
    """JpegEncoderHelper::compressImage through the oracle ("orc") or the image's libjpeg ("lj"); y: uint8 array holding
    the luma plane, uv: uint8 array holding U then V (V at chroma_stride*h/2) or None for a single plane -> bytes"""
    fn = load().orc_jpeg_encode if which == "orc" else load_libjpeg().lj_jpeg_encode
    ls = w if luma_stride is None else luma_stride
    cs = (w // 2 if chroma_stride is None else chroma_stride) if uv is not None else 0
    iccb = None if icc is None else np.frombuffer(icc, np.uint8)
    cap = 1 << 16
    while True:
        out = np.empty(cap, np.uint8)
        n = fn(y.ctypes.data, None if uv is None else uv.ctypes.data, w, h, ls, cs, quality,
               None if iccb is None else iccb.ctypes.data, 0 if iccb is None else iccb.size, out.ctypes.data, cap)
        if n <= cap:
            return out[:max(n, 0)].tobytes() if n >= 0 else None
        cap = int(n) + 16


def jpeg_decode(which, data):
    """JpegDecoderHelper::decompressImage(DECODE_TO_YCBCR) through the oracle ("orc") or the image's libjpeg ("lj"):
    -> (status, planes uint8 array, w, h, gray); status < 0 on failure"""
    fn = load().orc_jpeg_decode if which == "orc" else load_libjpeg().lj_jpeg_decode
    buf = np.frombuffer(data, np.uint8)
    w, h, g = C.c_int(), C.c_int(), C.c_int()
    cap = 1 << 16
    for _ in range(2):
        out = np.zeros(cap, np.uint8)
        n = fn(buf.ctypes.data, buf.size, out.ctypes.data, cap, C.byref(w), C.byref(h), C.byref(g))
        if n == -3:
            cap = w.value * h.value * 2 + 64
            continue
        break
    return int(n), (out[:n].copy() if n > 0 else None), w.value, h.value, g.value


def ycc420_to_rgba(planes, w, h):
    """decoded planes (Y, Cb, Cr as jpeg_decode returns them) -> (h, w, 4) uint8, libjpeg-turbo's DECODE_TO_RGBA arithmetic"""
    lib = load()
    lib.orc_ycc420_to_rgba.restype = C.c_int
    lib.orc_ycc420_to_rgba.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    planes = np.ascontiguousarray(planes)
    out = np.zeros((h, w, 4), np.uint8)
    base = planes.ctypes.data
    assert lib.orc_ycc420_to_rgba(base, base + w * h, base + w * h + (w // 2) * (h // 2), w, h, out.ctypes.data) == 0
    return out


def jpeg_coefficients(y, uv, w, h, quality, luma_stride=None, chroma_stride=None):
    lib = load()
    n = lib.orc_jpeg_block_count(w, h, 1 if uv is None else 0)
    coef = np.empty((n, 64), np.int16)
    ls = w if luma_stride is None else luma_stride
    cs = (w // 2 if chroma_stride is None else chroma_stride) if uv is not None else 0
    lib.orc_jpeg_coefficients(y.ctypes.data, None if uv is None else uv.ctypes.data, w, h, ls, cs, quality, coef.ctypes.data)
    return coef


# ------------------------------------------------------------------ numpy-level helpers

def _ptr(a):
    return a.ctypes.data if a is not None else None


def yuv420_image(buf, w, h, gamut, luma_stride=None, chroma_stride=None, chroma=None):
    """buf: uint8 array holding Y (luma_stride*h) then U,V planes unless `chroma` (separate
    uint8 array: U plane then V plane at chroma_stride*(h/2)) is given."""
    ls = luma_stride or w
    cs = chroma_stride or ls // 2
    cptr = _ptr(chroma) if chroma is not None else buf.ctypes.data + ls * h
    return Image(buf.ctypes.data, w, h, gamut, cptr, ls, cs, FMT_YUV420)


def p010_image(buf, w, h, gamut, luma_stride=None, chroma_stride=None, chroma=None):
    ls = luma_stride or w
    cs = chroma_stride or ls
    cptr = _ptr(chroma) if chroma is not None else buf.ctypes.data + ls * h * 2
    return Image(buf.ctypes.data, w, h, gamut, cptr, ls, cs, FMT_P010)


def map_image(buf, mw, mh):
    return Image(buf.ctypes.data, mw, mh, CG_UNSPECIFIED, None, mw, 0, FMT_MONOCHROME)


def out_bytes_per_image(fmt, w, h):
    return {OUT_HDR_LINEAR: 8, OUT_HDR_PQ: 4, OUT_HDR_HLG: 4, OUT_HDR_LINEAR_RGB_10BIT: 6}.get(fmt, 0) * w * h


def lcg_frame(w, h, seed):
    """SURVEY.md 8(d) synthetic pair: (p010 uint16[w*h*3/2], yuv uint8[w*h*3/2])."""
    lib = load()
    p010 = np.empty(w * h * 3 // 2, np.uint16)
    yuv = np.empty(w * h * 3 // 2, np.uint8)
    lib.orc_fill_lcg(p010.ctypes.data, yuv.ctypes.data, w, h, seed)
    return p010, yuv


def eval_transfer(fn, x, min_boost=1.0, max_boost=4.0, threads=8):
    """out[i] = f(x[i]) with the oracle's scalar functions; x float32 ndarray"""
    from concurrent.futures import ThreadPoolExecutor
    lib = load()
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    n = x.size
    step = (n + threads - 1) // threads

    def run(lo):
        hi = min(n, lo + step)
        lib.orc_eval_transfer(fn, x.ctypes.data + 4 * lo, out.ctypes.data + 4 * lo, hi - lo, min_boost, max_boost)

    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(run, range(0, n, step)))
    return out


def lut_table(which):
    """the oracle's static LUT `which` (0 srgbInv, 1 hlgInv, 2 pqInv, 4 hlg, 5 pq) as a float32 array"""
    n = C.c_size_t()
    p = load().orc_lut_table(which, C.byref(n))
    return np.ctypeslib.as_array(p, shape=(n.value,)).copy()


def gain_lut(lib_prefix, min_boost, max_boost, display_boost=None):
    """GainLUT(metadata) (display_boost None) or GainLUT(metadata, display_boost) as 1024 floats"""
    lib = load() if lib_prefix == "orc_" else load_ref()
    t = (C.c_float * 1024)()
    getattr(lib, lib_prefix + "gainLutBuild")(min_boost, max_boost, 0 if display_boost is None else 1,
                                              0.0 if display_boost is None else display_boost, t)
    return np.frombuffer(t, np.float32).copy()


def generate(lib_prefix, yuv_img, p010_img, tf, sdr_is_601=False, threads=0, stats=False, lut=False):
    lib = load() if lib_prefix == "orc_" else load_ref()
    mw, mh = yuv_img.width // 4, yuv_img.height // 4
    out = np.zeros(max(mw * mh, 1), np.uint8)
    md = Metadata()
    if stats:
        mm = (C.c_float * 2)()
        st = lib.orc_generateGainMapStats(C.byref(yuv_img), C.byref(p010_img), tf, C.byref(md),
                                          out.ctypes.data, int(sdr_is_601), threads, mm)
        return st, out[:mw * mh].reshape(mh, mw), md, (mm[0], mm[1])
    st = getattr(lib, lib_prefix + ("generateGainMapLUT" if lut else "generateGainMap"))(C.byref(yuv_img), C.byref(p010_img), tf,
                                                      C.byref(md), out.ctypes.data,
                                                      int(sdr_is_601), threads)
    return st, out[:mw * mh].reshape(mh, mw), md


def apply(lib_prefix, yuv_img, map_arr, md, fmt, max_display_boost, threads=0, lut=False):
    lib = load() if lib_prefix == "orc_" else load_ref()
    mh, mw = map_arr.shape
    m = map_image(map_arr, mw, mh)
    w, h = yuv_img.width, yuv_img.height
    out = np.zeros(max(out_bytes_per_image(fmt, w, h), 8), np.uint8)
    dest = Image(out.ctypes.data, 0, 0, CG_UNSPECIFIED, None, 0, 0, -1)
    st = getattr(lib, lib_prefix + ("applyGainMapLUT" if lut else "applyGainMap"))(C.byref(yuv_img), C.byref(m), C.byref(md), fmt,
                                                   max_display_boost, C.byref(dest), threads)
    return st, out[:out_bytes_per_image(fmt, w, h)], dest
