"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol include/uhdr_hip.h
declares, validates arguments with the reference's status codes, and REFUSES to compute without a GPU
(no CPU fallback, no route through oracle/)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def api():
    from libultrahdr_dev_amd import api as a
    a.load()
    return a


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "uhdr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(uhdr_hip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(api):
    names = _declared_functions()
    assert len(names) >= 12
    lib = C.CDLL(api.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libuhdr_hip.so does not export %s" % n
        assert n in api.SIGNATURES, "python binding lacks %s" % n
    assert lib.uhdr_hip_abi_version() == api.ABI_VERSION == 3
    out = subprocess.check_output(["nm", "-D", "--defined-only", api.LIB_PATH]).decode()
    exported = set(re.findall(r" T (uhdr_hip_\w+)", out))
    assert exported == set(names), exported ^ set(names)


def test_comm_library_exports_every_declared_symbol(api):
    """include/uhdr_hip_comm.h (the RCCL side, a library of its own): every declared entry point exported and bound, nothing else;
    argument validation needs no GPU"""
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "uhdr_hip_comm.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(uhdr_hip_comm_[a-z0-9_]+)\s*\(", text)))
    assert names == sorted(api.COMM_SIGNATURES) and len(names) == 5
    out = subprocess.check_output(["nm", "-D", "--defined-only", api.COMM_LIB_PATH]).decode()
    assert set(re.findall(r" T (uhdr_hip_\w+)", out)) == set(names)
    needed = subprocess.check_output(["readelf", "-d", api.COMM_LIB_PATH]).decode()
    assert "librccl.so" in needed
    assert "librccl" not in subprocess.check_output(["readelf", "-d", api.LIB_PATH]).decode()   # the pixel path does not depend on RCCL
    lib = api.load_comm()
    comm = C.c_void_p()
    ident = (C.c_char * api.COMM_ID_BYTES)()
    assert lib.uhdr_hip_comm_get_unique_id(None) == api.ERROR_BAD_PTR
    assert lib.uhdr_hip_comm_init(ident, 0, 0, 0, C.byref(comm)) == api.ERROR_BAD_PTR      # world < 1
    assert lib.uhdr_hip_comm_init(ident, 2, 2, 0, C.byref(comm)) == api.ERROR_BAD_PTR      # rank outside the world
    assert lib.uhdr_hip_comm_init(None, 1, 0, 0, C.byref(comm)) == api.ERROR_BAD_PTR and not comm.value
    assert lib.uhdr_hip_comm_allreduce_minmax(None, None, 0, None, None) == api.ERROR_BAD_PTR
    assert lib.uhdr_hip_comm_destroy(None) == api.ERROR_BAD_PTR


def test_cpp_multi_gpu_example_compiles_against_the_headers(tmp_path):
    """examples/multi_gpu_batch.cpp (the sharded step from a C++ host, one process per GPU) builds against include/*.h and links
    the two libraries; it runs only where there is a GPU (tests/test_gpu_comm.py)"""
    exe = str(tmp_path / "multi_gpu_batch")
    libdir = os.path.join(ROOT, "libultrahdr_dev_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
                           "-o", exe, os.path.join(ROOT, "examples", "multi_gpu_batch.cpp"), "-L" + libdir, "-luhdr_hip", "-luhdr_hip_comm",
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe, "--gpus", "0"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "bad arguments" in r.stderr


def test_shim_exports_reference_member_names():
    shim = os.path.join(ROOT, "libultrahdr_dev_amd", "libultrahdr_shim.so")
    assert os.path.exists(shim), "build() did not produce the C++ shim"
    out = subprocess.check_output(["nm", "-DC", "--defined-only", shim]).decode()
    for member in ("generateGainMap", "applyGainMap", "toneMap", "convertYuv"):
        assert "ultrahdr::UltraHdrHip::%s(" % member in out


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under libultrahdr_dev_amd/ or include/ may reference it"""
    for base in ("libultrahdr_dev_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    assert "liboracle" not in txt and "uhdr_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_no_gpu_means_no_compute(api):
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is for the CPU-only container")
    lib = api.load()
    assert lib.uhdr_hip_device_count() == 0
    assert lib.uhdr_hip_init(0) == api.ERROR_INSUFFICIENT_RESOURCE
    assert b"no CPU path" in lib.uhdr_hip_last_error()
    w, h = 16, 8
    yuv = np.zeros(w * h * 3 // 2, np.uint8)
    p010 = np.zeros(w * h * 3 // 2, np.uint16)
    gmap = np.full(8, 0x77, np.uint8)
    yi = api.yuv420_image(yuv.ctypes.data, w, h, api.CG_BT709)
    pi = api.p010_image(p010.ctypes.data, w, h, api.CG_BT2100)
    dest = api.out_image(gmap.ctypes.data)
    md = api.Metadata()
    rc = lib.uhdr_hip_generate_gainmap(C.byref(yi), C.byref(pi), api.TF_HLG, C.byref(md), C.byref(dest), 0, api.MEM_HOST, None)
    assert rc == api.ERROR_INSUFFICIENT_RESOURCE and (gmap == 0x77).all()
    out = np.full(w * h, 0x55555555, np.uint32)
    od = api.out_image(out.ctypes.data)
    mi = api.mono_image(gmap.ctypes.data, 4, 2)
    good = api.metadata(4.0)
    rc = lib.uhdr_hip_apply_gainmap(C.byref(yi), C.byref(mi), C.byref(good), api.OUTPUT_HDR_PQ, 4.0, C.byref(od), 0, api.MEM_HOST, None)
    assert rc == api.ERROR_INSUFFICIENT_RESOURCE and (out == 0x55555555).all()
    assert lib.uhdr_hip_tonemap(C.byref(pi), C.byref(yi), api.MEM_HOST, None) == api.ERROR_INSUFFICIENT_RESOURCE
    assert lib.uhdr_hip_convert_yuv(C.byref(yi), 0, 1, api.MEM_HOST, None) == api.ERROR_INSUFFICIENT_RESOURCE


def test_argument_validation_codes_without_device(api):
    """the reference's checks run before anything touches the device (ultrahdr.cpp:189-202,364-406,518-523;
    jpegr.cpp:1134-1147), so their status codes are observable on a CPU-only box"""
    lib = api.load()
    w, h = 16, 8
    yuv = np.zeros(w * h * 3 // 2, np.uint8)
    p010 = np.zeros(w * h * 3 // 2, np.uint16)
    buf = np.zeros(w * h * 8, np.uint8)
    yi = api.yuv420_image(yuv.ctypes.data, w, h, api.CG_BT709)
    pi = api.p010_image(p010.ctypes.data, w, h, api.CG_BT2100)
    dest = api.out_image(buf.ctypes.data)
    md = api.Metadata()
    gen = lambda y, p, tf, m, d: lib.uhdr_hip_generate_gainmap(y, p, tf, m, d, 0, api.MEM_HOST, None)
    assert gen(None, C.byref(pi), 1, C.byref(md), C.byref(dest)) == api.ERROR_BAD_PTR
    assert gen(C.byref(yi), None, 1, C.byref(md), C.byref(dest)) == api.ERROR_BAD_PTR
    assert gen(C.byref(yi), C.byref(pi), 1, None, C.byref(dest)) == api.ERROR_BAD_PTR
    assert gen(C.byref(yi), C.byref(pi), 1, C.byref(md), None) == api.ERROR_BAD_PTR
    nc = api.p010_image(p010.ctypes.data, w, h, 2); nc.chroma_data = None
    assert gen(C.byref(yi), C.byref(nc), 1, C.byref(md), C.byref(dest)) == api.ERROR_BAD_PTR
    p2 = api.p010_image(p010.ctypes.data, w, h + 2, 2)
    assert gen(C.byref(yi), C.byref(p2), 1, C.byref(md), C.byref(dest)) == api.ERROR_RESOLUTION_MISMATCH
    yu = api.yuv420_image(yuv.ctypes.data, w, h, api.CG_UNSPECIFIED)
    assert gen(C.byref(yu), C.byref(pi), 1, C.byref(md), C.byref(dest)) == api.ERROR_INVALID_COLORGAMUT
    # resolution mismatch wins over the gamut check, gamut over the transfer function (reference order)
    assert gen(C.byref(yu), C.byref(p2), 7, C.byref(md), C.byref(dest)) == api.ERROR_RESOLUTION_MISMATCH
    assert gen(C.byref(yu), C.byref(pi), 7, C.byref(md), C.byref(dest)) == api.ERROR_INVALID_COLORGAMUT
    assert gen(C.byref(yi), C.byref(pi), api.TF_SRGB, C.byref(md), C.byref(dest)) == api.ERROR_INVALID_TRANS_FUNC

    mi = api.mono_image(buf.ctypes.data, 4, 2)
    app = lambda y, m, meta, d: lib.uhdr_hip_apply_gainmap(y, m, meta, api.OUTPUT_HDR_PQ, 4.0, d, 0, api.MEM_HOST, None)
    good = api.metadata(4.0)
    assert app(None, C.byref(mi), C.byref(good), C.byref(dest)) == api.ERROR_BAD_PTR
    assert app(C.byref(yi), C.byref(mi), None, C.byref(dest)) == api.ERROR_BAD_PTR
    for mut in ("version", "gamma", "offsetSdr", "offsetHdr", "hdrCapacityMin", "hdrCapacityMax"):
        bad = api.metadata(4.0)
        setattr(bad, mut, b"2.0" if mut == "version" else 3.0)
        assert app(C.byref(yi), C.byref(mi), C.byref(bad), C.byref(dest)) == api.ERROR_BAD_METADATA, mut
    assert app(C.byref(yi), C.byref(api.mono_image(buf.ctypes.data, 5, 2)), C.byref(good), C.byref(dest)) == api.ERROR_UNSUPPORTED_MAP_SCALE_FACTOR
    assert app(C.byref(yi), C.byref(api.mono_image(buf.ctypes.data, 4, 4)), C.byref(good), C.byref(dest)) == api.ERROR_UNSUPPORTED_MAP_SCALE_FACTOR
    # bad metadata is reported before a bad scale factor
    bad = api.metadata(4.0); bad.gamma = 2.0
    assert app(C.byref(yi), C.byref(api.mono_image(buf.ctypes.data, 5, 2)), C.byref(bad), C.byref(dest)) == api.ERROR_BAD_METADATA

    assert lib.uhdr_hip_tonemap(None, C.byref(yi), api.MEM_HOST, None) == api.ERROR_BAD_PTR
    y2 = api.yuv420_image(yuv.ctypes.data, w + 2, h, 0)
    assert lib.uhdr_hip_tonemap(C.byref(pi), C.byref(y2), api.MEM_HOST, None) == api.ERROR_RESOLUTION_MISMATCH
    assert lib.uhdr_hip_convert_yuv(None, 0, 1, api.MEM_HOST, None) == api.ERROR_BAD_PTR
    assert lib.uhdr_hip_convert_yuv(C.byref(yi), -1, 1, api.MEM_HOST, None) == api.ERROR_INVALID_COLORGAMUT
    assert lib.uhdr_hip_convert_yuv(C.byref(yi), 0, -1, api.MEM_HOST, None) == api.ERROR_INVALID_COLORGAMUT
    assert lib.uhdr_hip_convert_yuv(C.byref(yi), 2, 2, api.MEM_HOST, None) == api.NO_ERROR   # same encoding: no-op


def test_idw_tables_equal_the_oracle(api, orc):
    """the Shepard IDW weights the apply kernels use (host-computed once per scale) are bit-identical to
    the reference's fillShepardsIDW (gainmapmath.cpp:69-110)"""
    lib, L = api.load(), orc.load()
    for scale in (1, 2, 3, 4, 5, 6, 8):
        n = scale * scale * 4
        got = (C.c_float * (4 * n))()
        assert lib.uhdr_hip_idw_tables(scale, got) == 0
        for t, (incR, incB) in enumerate(((1, 1), (0, 1), (1, 0), (0, 0))):
            want = (C.c_float * n)()
            L.orc_fillShepardsIDW(want, scale, incR, incB)
            assert list(got)[t * n:(t + 1) * n] == list(want), (scale, t)


def test_synthetic_frames_match_the_survey_lcg(orc):
    """libultrahdr_dev_amd.synth (jump-ahead LCG, used by bench.py on the GPU) == SURVEY 8(d) serial LCG"""
    from libultrahdr_dev_amd import synth
    for (w, h, seed) in ((64, 32, 1234), (40, 24, 1299)):
        p, y = synth.lcg_frame(w, h, seed, device="cpu")
        op, oy = orc.lcg_frame(w, h, seed)
        assert np.array_equal(p.numpy().view(np.uint16), op) and np.array_equal(y.numpy(), oy)


def test_graft_entry_build_runs():
    """the driver's "does it build" check: __graft_entry__.build() compiles the library, the shim and the checkers and loads the library"""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    g = importlib.import_module("__graft_entry__")
    g.build()


def test_jpeg_header_refuses_a_huffman_table_that_is_no_prefix_code(api, orc):
    """parse_header sees the untrusted file before the device does (no GPU needed to refuse it): a DHT whose counts put more codes
    on a length than its bits hold is JERR_BAD_HUFF_TABLE in libjpeg (jdhuff.c, jpeg_make_d_derived_tbl) and UNKNOWN_ERROR here --
    the canonical long-code lookup of the device decoder relies on the tables being prefix codes"""
    lib = api.load()
    rng = np.random.default_rng(5)
    w, h = 64, 48
    y = rng.integers(0, 256, w * h, dtype=np.uint8)
    uv = rng.integers(0, 256, w * h // 2, dtype=np.uint8)
    good = bytearray(orc.jpeg_encode("orc", y, uv, w, h, 90))
    out = np.zeros(w * h * 3 // 2, np.uint8)
    desc = api.Image()

    def decode(data):
        buf = np.frombuffer(bytes(data), np.uint8)
        return lib.uhdr_hip_jpeg_decode(C.c_void_p(buf.ctypes.data), buf.size, C.c_void_p(out.ctypes.data), out.size, C.byref(desc), api.MEM_HOST, None)

    assert orc.jpeg_decode("lj", bytes(good))[0] > 0
    at = bytes(good).index(b"\xff\xc4")      # first DHT: marker, length, Tc/Th, 16 counts, symbols
    counts = at + 5
    bad = bytearray(good)
    donor = max(range(16), key=lambda k: bad[counts + k])
    assert bad[counts + donor] >= 3 and donor != 0
    bad[counts + donor] -= 3                 # same number of symbols (the segment stays well-formed) ...
    bad[counts + 0] += 3                     # ... but three 1-bit codes
    assert orc.jpeg_decode("lj", bytes(bad))[0] < 0, "libjpeg accepts the table"
    assert decode(bad) == api.UNKNOWN_ERROR


_OOM_PROBE = r'''
import ctypes as C, io, os, resource, sys
import numpy as np
from PIL import Image
sys.path.insert(0, %r)
from libultrahdr_dev_amd import api
lib = api.load()
# a progressive 4:2:0 file whose coefficient array (4096 x 4096: 50 MB of int16) the header parser allocates on the host
b = io.BytesIO()
Image.fromarray(np.full((4096, 4096, 3), 128, np.uint8)).save(b, "JPEG", quality=50, progressive=True, subsampling="4:2:0")
data = np.frombuffer(b.getvalue(), np.uint8).copy()
del b
desc = api.Image()
assert lib.uhdr_hip_jpeg_decode(C.c_void_p(data.ctypes.data), data.size, None, 0, C.byref(desc), api.MEM_DEVICE, None) == api.ERROR_INSUFFICIENT_RESOURCE
assert (desc.width, desc.height) == (4096, 4096)          # the size probe, with memory to spare: the header parsed, desc filled
vm = int(open("/proc/self/statm").read().split()[0]) * os.sysconf("SC_PAGE_SIZE")
resource.setrlimit(resource.RLIMIT_AS, (vm + (24 << 20), resource.RLIM_INFINITY))   # 24 MiB of headroom: the 50 MB array cannot be had
sentinel = api.Image(C.c_void_p(0x1234), 77, 99, 1, None, 5, 6, 2)
desc = api.Image(C.c_void_p(0x1234), 77, 99, 1, None, 5, 6, 2)
rc = lib.uhdr_hip_jpeg_decode(C.c_void_p(data.ctypes.data), data.size, None, 0, C.byref(desc), api.MEM_DEVICE, None)
print("decode", rc, desc.width, desc.height)
desc2 = api.Image(C.c_void_p(0x1234), 77, 99, 1, None, 5, 6, 2)
rc2 = lib.uhdr_hip_jpeg_decode_rgba(C.c_void_p(data.ctypes.data), data.size, None, 0, C.byref(desc2), api.MEM_DEVICE, None)
print("rgba", rc2, desc2.width, desc2.height)
nb, w, h, g = C.c_size_t(7), C.c_int(7), C.c_int(7), C.c_int(7)
rc3 = lib.uhdr_hip_jpeg_progressive_coefficients(C.c_void_p(data.ctypes.data), data.size, None, 0, C.byref(nb), C.byref(w), C.byref(h), C.byref(g))
print("coef", rc3, nb.value, w.value, h.value)
p010 = np.zeros(64 * 64 * 3 // 2, np.uint16)
pi = api.p010_image(p010.ctypes.data, 64, 64, api.CG_BT2100)
out, n = np.zeros(1 << 16, np.uint8), C.c_size_t(7)
rc4 = lib.uhdr_hip_jpegr_encode_api3(C.byref(pi), C.c_void_p(data.ctypes.data), data.size, api.CG_BT709, api.TF_HLG, C.c_void_p(out.ctypes.data), out.size, C.byref(n), api.MEM_HOST, None)
print("api3", rc4, n.value)
'''


def test_host_allocation_failure_is_never_the_size_probe_answer(api):
    """ERROR_INSUFFICIENT_RESOURCE from uhdr_hip_jpeg_decode with out == NULL is the size probe -- the header
    parsed and *desc holds the size -- and four callers (decode_rgba, encodeJPEGR API-3, the shim's two decompressImage paths) trust
    desc on it.  A std::bad_alloc inside the header parser (a progressive file's coefficient array) must therefore come back as
    another status and leave no size behind.  Forced here with RLIMIT_AS in a child process; no GPU involved (host parsing)."""
    pytest.importorskip("PIL")
    import sys
    r = subprocess.run([sys.executable, "-c", _OOM_PROBE % ROOT], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    got = dict((l.split()[0], [int(x) for x in l.split()[1:]]) for l in r.stdout.splitlines() if l.split() and l.split()[0] in ("decode", "rgba", "coef", "api3"))
    assert got["decode"] == [api.UNKNOWN_ERROR, 0, 0], got            # not the probe status; desc zeroed
    assert got["rgba"][0] == api.UNKNOWN_ERROR and got["rgba"][1:] == [77, 99], got   # the caller's desc untouched
    assert got["coef"][0] == api.UNKNOWN_ERROR and got["coef"][1] == 7, got
    assert got["api3"][0] == api.ERROR_DECODE_ERROR, got
