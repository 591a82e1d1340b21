"""ctypes binding of include/uhdr_hip.h (the drop-in C-ABI).  No compute happens in Python.

Names follow the reference (lib/include/ultrahdr/ultrahdr.h): images are described by the fields
of ``ultrahdr_uncompressed_struct``, metadata by ``ultrahdr_metadata_struct``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UHDR_HIP_LIB") or os.path.join(_HERE, "libuhdr_hip.so")  # env override: A/B builds

# enum values: ultrahdr.h:36-120
CG_UNSPECIFIED, CG_BT709, CG_P3, CG_BT2100 = -1, 0, 1, 2
TF_LINEAR, TF_HLG, TF_PQ, TF_SRGB = 0, 1, 2, 3
OUTPUT_SDR, OUTPUT_HDR_LINEAR, OUTPUT_HDR_PQ, OUTPUT_HDR_HLG, OUTPUT_HDR_LINEAR_RGB_10BIT = 0, 1, 2, 3, 4
PIX_FMT_P010, PIX_FMT_YUV420, PIX_FMT_MONOCHROME = 0, 1, 2
NO_ERROR, UNKNOWN_ERROR = 0, -1
ERROR_BAD_PTR, ERROR_INVALID_COLORGAMUT, ERROR_INVALID_TRANS_FUNC = -10001, -10003, -10005
ERROR_RESOLUTION_MISMATCH, ERROR_BAD_METADATA = -10006, -10010
ERROR_INVALID_CROPPING_PARAMETERS, ERROR_UNSUPPORTED_FEATURE = -10011, -30000
ERROR_UNSUPPORTED_MAP_SCALE_FACTOR, ERROR_INSUFFICIENT_RESOURCE = -20008, -20009
ERROR_UNSUPPORTED_WIDTH_HEIGHT, ERROR_INVALID_STRIDE, ERROR_INVALID_QUALITY_FACTOR = -10002, -10004, -10007
ERROR_INVALID_DISPLAY_BOOST, ERROR_INVALID_OUTPUT_FORMAT = -10008, -10009
ERROR_ENCODE_ERROR, ERROR_DECODE_ERROR, ERROR_GAIN_MAP_IMAGE_NOT_FOUND, ERROR_METADATA_ERROR = -20001, -20002, -20003, -20005
ERROR_NO_IMAGES_FOUND, ERROR_MULTIPLE_EXIFS_RECEIVED = -20006, -20007
MEM_HOST, MEM_DEVICE = 0, 1
APPLY_FAST, APPLY_EXACT, APPLY_LUT, APPLY_EXACT_UNFILTERED = 0, 1, 2, 3
GENERATE_EXACT, GENERATE_LUT, GENERATE_UNFILTERED = 0, 1, 2
ABI_VERSION = 3
FLT_MAX = 3.4028234663852886e38


class Image(C.Structure):
    """uhdr_hip_image_t == ultrahdr_uncompressed_struct (ultrahdr.h:152-181); strides in pixels."""
    _fields_ = [("data", C.c_void_p), ("width", C.c_size_t), ("height", C.c_size_t),
                ("colorGamut", C.c_int32), ("chroma_data", C.c_void_p),
                ("luma_stride", C.c_size_t), ("chroma_stride", C.c_size_t),
                ("pixelFormat", C.c_int32)]


class Effect(C.Structure):
    """uhdr_hip_effect_t: type 0 crop(a=left,b=right,c=top,d=bottom) 1 mirror(a=dir) 2 rotate(a=degrees) 3 resize(a=w,b=h)"""
    _fields_ = [("type", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("c", C.c_int32), ("d", C.c_int32)]


class JpegInfo(C.Structure):
    """uhdr_hip_jpeg_info_t: one image of a JPEG/R file, ranges into the file"""
    _fields_ = [(n, C.c_size_t) for n in ("offset", "size", "width", "height", "icc_offset", "icc_size", "exif_offset", "exif_size", "xmp_offset", "xmp_size")]


class Metadata(C.Structure):
    """uhdr_hip_metadata_t == ultrahdr_metadata_struct (ultrahdr.h:129-147)."""
    _fields_ = [("version", C.c_char * 8), ("maxContentBoost", C.c_float),
                ("minContentBoost", C.c_float), ("gamma", C.c_float), ("offsetSdr", C.c_float),
                ("offsetHdr", C.c_float), ("hdrCapacityMin", C.c_float),
                ("hdrCapacityMax", C.c_float)]


# every symbol include/uhdr_hip.h declares, with its signature
_IP, _MP = C.POINTER(Image), C.POINTER(Metadata)
SIGNATURES = {
    "uhdr_hip_abi_version": (C.c_int, []),
    "uhdr_hip_device_count": (C.c_int, []),
    "uhdr_hip_init": (C.c_int, [C.c_int]),
    "uhdr_hip_shutdown": (C.c_int, []),
    "uhdr_hip_stream_reserve": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_int]),
    "uhdr_hip_stream_release": (C.c_int, [C.c_void_p]),
    "uhdr_hip_last_error": (C.c_char_p, []),
    "uhdr_hip_generate_gainmap": (C.c_int, [_IP, _IP, C.c_int, _MP, _IP, C.c_int, C.c_int, C.c_void_p]),
    "uhdr_hip_apply_gainmap": (C.c_int, [_IP, _IP, _MP, C.c_int, C.c_float, _IP, C.c_int, C.c_int, C.c_void_p]),
    "uhdr_hip_tonemap": (C.c_int, [_IP, _IP, C.c_int, C.c_void_p]),
    "uhdr_hip_convert_yuv": (C.c_int, [_IP, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "uhdr_hip_tonemap_batch": (C.c_int, [C.c_int, _IP, _IP, C.c_void_p]),
    "uhdr_hip_convert_yuv_batch": (C.c_int, [C.c_int, _IP, C.c_int, C.c_int, C.c_void_p]),
    "uhdr_hip_generate_gainmap_batch": (C.c_int, [C.c_int, _IP, _IP, C.c_int, _MP, _IP, C.c_int, C.c_void_p, C.c_void_p]),
    "uhdr_hip_apply_gainmap_batch": (C.c_int, [C.c_int, _IP, _IP, _MP, C.c_int, C.c_float, _IP, C.c_int, C.c_void_p]),
    "uhdr_hip_mem_pool_create": (C.c_int, [C.c_int, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p)]),
    "uhdr_hip_mem_pool_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "uhdr_hip_mem_pool_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "uhdr_hip_mem_pool_trim": (C.c_int, [C.c_void_p]),
    "uhdr_hip_mem_pool_destroy": (C.c_int, [C.c_void_p]),
    "uhdr_hip_mem_pool_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "uhdr_hip_crop": (C.c_int, [_IP, C.c_int, C.c_int, C.c_int, C.c_int, _IP, C.c_int, C.c_void_p]),
    "uhdr_hip_mirror": (C.c_int, [_IP, C.c_int, _IP, C.c_int, C.c_void_p]),
    "uhdr_hip_rotate": (C.c_int, [_IP, C.c_int, _IP, C.c_int, C.c_void_p]),
    "uhdr_hip_resize": (C.c_int, [_IP, C.c_int, C.c_int, _IP, C.c_int, C.c_void_p]),
    "uhdr_hip_generate_gainmap_ex": (C.c_int, [_IP, _IP, C.c_int, _MP, _IP, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "uhdr_hip_generate_gainmap_batch_ex": (C.c_int, [C.c_int, _IP, _IP, C.c_int, _MP, _IP, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "uhdr_hip_add_effects": (C.c_int, [_IP, C.c_void_p, C.c_int, _IP, C.c_int, C.c_void_p]),
    "uhdr_hip_jpeg_progressive_coefficients": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int),
                                                         C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "uhdr_hip_jpeg_encode": (C.c_int, [_IP, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_int, C.c_void_p]),
    "uhdr_hip_jpeg_decode": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, _IP, C.c_int, C.c_void_p]),
    "uhdr_hip_jpegr_decode": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_float, C.c_void_p, C.c_size_t, _IP, _MP, C.c_int, C.c_int, C.c_void_p]),
    "uhdr_hip_jpegr_append_gainmap": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, _MP,
                                                C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "uhdr_hip_icc_profile": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "uhdr_hip_jpegr_encode_api0": (C.c_int, [_IP, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_int, C.c_void_p]),
    "uhdr_hip_jpegr_encode_api1": (C.c_int, [_IP, _IP, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_int,
                                             C.c_void_p]),
    "uhdr_hip_jpegr_encode_api2": (C.c_int, [_IP, _IP, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_int,
                                             C.c_void_p]),
    "uhdr_hip_jpegr_encode_api3": (C.c_int, [_IP, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_int, C.c_void_p]),
    "uhdr_hip_jpegr_encode_api4": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, _MP, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "uhdr_hip_jpegr_encode_apix": (C.c_int, [_IP, _IP, _MP, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_int, C.c_void_p]),
    "uhdr_hip_jpeg_decode_rgba": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, _IP, C.c_int, C.c_void_p]),
    "uhdr_hip_jpegr_decode_batch": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_float, C.POINTER(C.c_void_p),
                                              C.POINTER(C.c_size_t), _IP, _MP, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p]),
    "uhdr_hip_jpegr_metadata": (C.c_int, [C.c_void_p, C.c_size_t, _MP]),
    "uhdr_hip_jpegr_info": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(JpegInfo), C.POINTER(JpegInfo)]),
    "uhdr_hip_lut_table": (C.c_int, [C.c_int, C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_size_t)]),
    "uhdr_hip_gain_lut": (C.c_int, [_MP, C.c_int, C.c_float, C.POINTER(C.c_float)]),
    "uhdr_hip_idw_tables": (C.c_int, [C.c_int, C.POINTER(C.c_float)]),
    "uhdr_hip_eval_transfer": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_void_p]),
    "uhdr_hip_synth_lcg_frame": (C.c_int, [C.c_size_t, C.c_size_t, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None


def load():
    """dlopen libuhdr_hip.so; raises if it has not been built (no fallback of any kind)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libuhdr_hip.so is missing: build the HIP extension first "
                "(python -m libultrahdr_dev_amd.build).  There is no CPU path.")
        # In a PyTorch process the library must bind to the HIP runtime torch bundles
        # (torch/lib/libamdhip64.so), otherwise two HIP/HSA runtimes end up in one process and the
        # second one sees no device.  Importing torch first makes the loader resolve our
        # libamdhip64.so.7 dependency to the copy that is already mapped.
        try:
            import torch  # noqa: F401  (plumbing: device memory + streams for tests/bench)
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.uhdr_hip_abi_version() != ABI_VERSION:
            raise ImportError("libuhdr_hip.so ABI version mismatch")
        _lib = lib
    return _lib


# include/uhdr_hip_comm.h -- libuhdr_hip_comm.so: the path's one exchange between GPUs (RCCL), apart from the pixel library
COMM_LIB_PATH = os.path.join(_HERE, "libuhdr_hip_comm.so")
COMM_ID_BYTES = 128
COMM_SIGNATURES = {
    "uhdr_hip_comm_get_unique_id": (C.c_int, [C.c_void_p]),
    "uhdr_hip_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "uhdr_hip_comm_world": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "uhdr_hip_comm_allreduce_minmax": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "uhdr_hip_comm_destroy": (C.c_int, [C.c_void_p]),
}
_comm_lib = None


def load_comm():
    """dlopen libuhdr_hip_comm.so (links librccl.so.1: in a PyTorch process the loader hands it the copy torch has mapped)."""
    global _comm_lib
    if _comm_lib is None:
        if not os.path.exists(COMM_LIB_PATH):
            raise ImportError("libuhdr_hip_comm.so is missing: python -m libultrahdr_dev_amd.build")
        try:
            import torch  # noqa: F401  (same reason as in load(): one HIP runtime, one RCCL per process)
        except ImportError:
            pass
        lib = C.CDLL(COMM_LIB_PATH)
        for name, (res, args) in COMM_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _comm_lib = lib
    return _comm_lib


class UhdrHipError(RuntimeError):
    pass


_inited = set()


def init(device=0):
    """uhdr_hip_init(device); raises when there is no usable GPU."""
    lib = load()
    if device not in _inited:
        rc = lib.uhdr_hip_init(device)
        if rc != NO_ERROR:
            raise UhdrHipError("uhdr_hip_init(%d) -> %d: %s" % (device, rc, lib.uhdr_hip_last_error().decode()))
        _inited.add(device)
    return lib


# ---- descriptor helpers (device or host pointers; nothing is copied) --------------------------------

def yuv420_image(ptr, w, h, gamut, luma_stride=None, chroma_stride=None, chroma_ptr=None):
    ls = w if luma_stride is None else luma_stride
    cs = ls // 2 if chroma_stride is None else chroma_stride
    cp = ptr + ls * h if chroma_ptr is None else chroma_ptr
    return Image(ptr, w, h, gamut, cp, ls, cs, PIX_FMT_YUV420)


def p010_image(ptr, w, h, gamut, luma_stride=None, chroma_stride=None, chroma_ptr=None):
    ls = w if luma_stride is None else luma_stride
    cs = ls if chroma_stride is None else chroma_stride
    cp = ptr + ls * h * 2 if chroma_ptr is None else chroma_ptr
    return Image(ptr, w, h, gamut, cp, ls, cs, PIX_FMT_P010)


def mono_image(ptr, w, h):
    return Image(ptr, w, h, CG_UNSPECIFIED, None, w, 0, PIX_FMT_MONOCHROME)


def out_image(ptr):
    return Image(ptr, 0, 0, CG_UNSPECIFIED, None, 0, 0, -1)


def output_bytes(fmt, w, h):
    """applyGainMap's output size; OUTPUT_SDR: the RGBA8888 rendition decodeJPEGR returns (applyGainMap itself writes nothing for it)"""
    return {OUTPUT_HDR_LINEAR: 8, OUTPUT_HDR_PQ: 4, OUTPUT_HDR_HLG: 4,
            OUTPUT_HDR_LINEAR_RGB_10BIT: 6, OUTPUT_SDR: 4}.get(fmt, 0) * w * h


def image_array(images):
    arr = (Image * len(images))()
    for i, im in enumerate(images):
        arr[i] = im
    return arr


def metadata(max_boost, min_boost=1.0, version=b"1.0"):
    return Metadata(version, float(max_boost), float(min_boost), 1.0, 0.0, 0.0, float(min_boost), float(max_boost))


class MemPool:
    """uhdr_hip_mem_pool_* (include/uhdr_hip.h, "where resident images lie in device memory"): device memory taken as chunks, every
    allocation spread evenly over the pool's free chunks.  `tensor(n)` returns a torch uint8 tensor over such an allocation (the pool
    must outlive it)."""

    def __init__(self, device, nbytes, chunk_bytes=0):
        self.lib, self.device = init(device), device
        self.handle = C.c_void_p()
        rc = self.lib.uhdr_hip_mem_pool_create(device, nbytes, chunk_bytes, C.byref(self.handle))
        if rc != NO_ERROR:
            raise MemoryError("uhdr_hip_mem_pool_create(%d bytes): status %d (%s)" % (nbytes, rc, self.lib.uhdr_hip_last_error().decode()))

    def alloc(self, nbytes):
        p = C.c_void_p()
        rc = self.lib.uhdr_hip_mem_pool_alloc(self.handle, nbytes, C.byref(p))
        if rc != NO_ERROR:
            raise MemoryError("uhdr_hip_mem_pool_alloc(%d bytes): status %d" % (nbytes, rc))
        return p.value

    def tensor(self, nbytes):
        import torch

        class _Raw:   # (torch reads the CUDA array interface; the object keeps nothing alive: the pool owns the memory)
            def __init__(self, ptr, n):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}
        return torch.as_tensor(_Raw(self.alloc(nbytes), nbytes), device=torch.device("cuda", self.device))

    def free(self, ptr):
        return self.lib.uhdr_hip_mem_pool_free(self.handle, C.c_void_p(ptr))

    def trim(self):
        return self.lib.uhdr_hip_mem_pool_trim(self.handle)

    def stats(self):
        a, b = C.c_size_t(), C.c_size_t()
        assert self.lib.uhdr_hip_mem_pool_stats(self.handle, C.byref(a), C.byref(b)) == NO_ERROR
        return a.value, b.value

    def destroy(self):
        if self.handle:
            rc = self.lib.uhdr_hip_mem_pool_destroy(self.handle)
            self.handle = C.c_void_p()
            return rc
        return NO_ERROR
