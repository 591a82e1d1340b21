"""-m gpu: the device JPEG encoder (uhdr_hip_jpeg_encode = JpegEncoderHelper::compressImage,
lib/src/jpegencoderhelper.cpp:39-283) against the CPU checker oracle/jpeg_oracle.c, whole files, BYTE FOR BYTE.
The checker itself is pinned to the image's libjpeg builds by tests/test_jpeg_oracle.py (CPU)."""
import ctypes as C

import numpy as np
import pytest

from tests.test_jpeg_oracle import SIZES, _content, _planes

pytestmark = pytest.mark.gpu


def _gpu_encode(lib, hip, yb, ub, w, h, q, ls, cs, device, icc=None, cap=None):
    from tests.gpu_util import dev_empty, stream_ptr, to_dev, to_host
    gray = ub is None
    n = C.c_size_t()
    iccp = None if icc is None else C.c_char_p(icc)
    iccn = 0 if icc is None else len(icc)
    if device:
        dy = to_dev(yb)
        du = None if gray else to_dev(ub)
        img = hip.Image(dy.data_ptr(), w, h, hip.CG_UNSPECIFIED, None if gray else du.data_ptr(), ls, cs,
                        hip.PIX_FMT_MONOCHROME if gray else hip.PIX_FMT_YUV420)
        cap = cap if cap is not None else w * h * 4 + 4096
        dout = dev_empty(cap, 0xCD)
        rc = lib.uhdr_hip_jpeg_encode(C.byref(img), q, iccp, iccn, C.c_void_p(dout.data_ptr()), cap, C.byref(n), hip.MEM_DEVICE, stream_ptr())
        return rc, n.value, (to_host(dout, min(n.value, cap)).tobytes() if rc == 0 else None)
    img = hip.Image(yb.ctypes.data, w, h, hip.CG_UNSPECIFIED, None if gray else ub.ctypes.data, ls, cs,
                    hip.PIX_FMT_MONOCHROME if gray else hip.PIX_FMT_YUV420)
    cap = cap if cap is not None else w * h * 4 + 4096
    out = np.full(cap, 0xCD, np.uint8)
    rc = lib.uhdr_hip_jpeg_encode(C.byref(img), q, iccp, iccn, C.c_void_p(out.ctypes.data), cap, C.byref(n), hip.MEM_HOST, None)
    return rc, n.value, (out[:n.value].tobytes() if rc == 0 else None)


@pytest.mark.parametrize("kind", ["smooth", "noise", "extreme", "flat"])
@pytest.mark.parametrize("device", [True, False])
def test_encoder_is_byte_exact_for_every_size_class(hip, orc, kind, device):
    lib = hip.load()
    rng = np.random.RandomState(len(kind) + int(device))
    for w, h in SIZES:
        y, u, v = _content(kind, w, h, rng)
        aw, acw = (w + 15) // 16 * 16, (w // 2 + 7) // 8 * 8
        for ls, cs in ((w, w // 2), (aw, acw), (aw + 16, acw + 8), (w + 2, w // 2 + 1)):
            yb, ub = _planes(y, u, v, ls, cs, rng)
            for q in (90, 85, 50, 20, 1, 100) if (w, h) in ((64, 48), (40, 24)) else (85,):
                rc, n, got = _gpu_encode(lib, hip, yb, ub, w, h, q, ls, cs, device)
                want = orc.jpeg_encode("orc", yb, ub, w, h, q, ls, cs)
                assert rc == 0 and n == len(want) and got == want, ("yuv420", kind, w, h, ls, cs, q, n, len(want))
                rc, n, got = _gpu_encode(lib, hip, yb, None, w, h, q, ls, 0, device)
                want = orc.jpeg_encode("orc", yb, None, w, h, q, ls)
                assert rc == 0 and got == want, ("plane", kind, w, h, ls, q, n, len(want))


def test_icc_segment_capacity_and_argument_errors(hip, orc):
    lib = hip.load()
    rng = np.random.RandomState(9)
    w, h = 64, 48
    y, u, v = _content("smooth", w, h, rng)
    yb, ub = _planes(y, u, v, w, w // 2, rng)
    icc = bytes(range(256)) * 2
    for device in (True, False):
        rc, n, got = _gpu_encode(lib, hip, yb, ub, w, h, 90, w, w // 2, device, icc=icc)
        assert rc == 0 and got == orc.jpeg_encode("orc", yb, ub, w, h, 90, icc=icc)
        rc, n, _ = _gpu_encode(lib, hip, yb, ub, w, h, 90, w, w // 2, device, cap=700)      # header fits, data does not
        assert rc == hip.ERROR_INSUFFICIENT_RESOURCE and n == len(orc.jpeg_encode("orc", yb, ub, w, h, 90))
        rc, n, _ = _gpu_encode(lib, hip, yb, ub, w, h, 90, w, w // 2, device, cap=16)       # not even the header
        assert rc == hip.ERROR_INSUFFICIENT_RESOURCE and n == len(orc.jpeg_encode("orc", yb, ub, w, h, 90))
    n = C.c_size_t()
    img = hip.Image(yb.ctypes.data, 63, 48, -1, ub.ctypes.data, 64, 32, hip.PIX_FMT_YUV420)
    out = np.zeros(1 << 16, np.uint8)
    assert lib.uhdr_hip_jpeg_encode(C.byref(img), 90, None, 0, C.c_void_p(out.ctypes.data), out.size, C.byref(n), hip.MEM_HOST, None) == hip.ERROR_RESOLUTION_MISMATCH
    assert lib.uhdr_hip_jpeg_encode(None, 90, None, 0, C.c_void_p(out.ctypes.data), out.size, C.byref(n), hip.MEM_HOST, None) == hip.ERROR_BAD_PTR
    img = hip.Image(yb.ctypes.data, 64, 48, -1, None, 64, 32, hip.PIX_FMT_YUV420)
    assert lib.uhdr_hip_jpeg_encode(C.byref(img), 90, None, 0, C.c_void_p(out.ctypes.data), out.size, C.byref(n), hip.MEM_HOST, None) == hip.ERROR_BAD_PTR


@pytest.mark.parametrize("q", [95, 85])
def test_4k_frame_and_its_gain_map(hip, orc, q):
    """configs[2]-sized content: a 3840x2160 YUV420 frame (base image, quality 95 in the reference's API-0/1) and its
    960x540 gain map (quality 85, jpegr.cpp:294-297), both produced on the device and compressed there"""
    from tests.gpu_util import gpu_generate
    from tests.test_gpu_parity import smooth_frame
    lib = hip.load()
    w, h = 3840, 2160
    p010, yuv = smooth_frame(w, h, 5)
    rc, n, got = _gpu_encode(lib, hip, yuv[:w * h], yuv[w * h:], w, h, q, w, w // 2, True)
    want = orc.jpeg_encode("orc", yuv[:w * h], yuv[w * h:], w, h, q)
    assert rc == 0 and got == want, (n, len(want))
    from tests.gpu_util import to_dev
    dp, dy = to_dev(p010), to_dev(yuv)
    st, gmap, _, _ = gpu_generate(lib, hip.yuv420_image(dy.data_ptr(), w, h, hip.CG_BT709), hip.p010_image(dp.data_ptr(), w, h, hip.CG_BT2100), hip.TF_HLG)
    assert st == 0
    gm = np.ascontiguousarray(gmap.reshape(-1))
    rc, n, got = _gpu_encode(lib, hip, gm, None, w // 4, h // 4, q, w // 4, 0, True)
    assert rc == 0 and got == orc.jpeg_encode("orc", gm, None, w // 4, h // 4, q)


# ---------------------------------------------------------------------------------------------------------------------
# decoder: uhdr_hip_jpeg_decode = JpegDecoderHelper::decompressImage(..., DECODE_TO_YCBCR)
# ---------------------------------------------------------------------------------------------------------------------
def _gpu_decode(lib, hip, data, device, cap=None):
    from tests.gpu_util import dev_empty, stream_ptr, to_host
    buf = np.frombuffer(data, np.uint8)
    desc = hip.Image()
    probe = lib.uhdr_hip_jpeg_decode(C.c_void_p(buf.ctypes.data), buf.size, None, 0, C.byref(desc), hip.MEM_HOST, None)
    if probe != hip.ERROR_INSUFFICIENT_RESOURCE:
        return probe, None, desc
    need = desc.width * desc.height * (1 if desc.pixelFormat == hip.PIX_FMT_MONOCHROME else 3) // (1 if desc.pixelFormat == hip.PIX_FMT_MONOCHROME else 2)
    if desc.pixelFormat != hip.PIX_FMT_MONOCHROME:
        need = desc.width * desc.height + 2 * (desc.width * desc.height // 4)
    cap = need if cap is None else cap
    if device:
        dout = dev_empty(cap + 64, 0xCD)
        rc = lib.uhdr_hip_jpeg_decode(C.c_void_p(buf.ctypes.data), buf.size, C.c_void_p(dout.data_ptr()), cap, C.byref(desc), hip.MEM_DEVICE, stream_ptr())
        return rc, (to_host(dout, need).copy() if rc == 0 else None), desc
    out = np.full(cap + 64, 0xCD, np.uint8)
    rc = lib.uhdr_hip_jpeg_decode(C.c_void_p(buf.ctypes.data), buf.size, C.c_void_p(out.ctypes.data), cap, C.byref(desc), hip.MEM_HOST, None)
    return rc, (out[:need].copy() if rc == 0 else None), desc


@pytest.mark.parametrize("device", [True, False])
def test_decoder_is_byte_exact_on_the_corpus(hip, orc, tmp_path, device):
    from tests.test_jpeg_oracle import jpeg_corpus
    lib = hip.load()
    corpus, extra = jpeg_corpus(orc, tmp_path)
    for name, data in corpus + [(k, extra[k]) for k in ("opt", "gray_opt", "rst", "rst_rows") if k in extra]:
        rc, got, desc = _gpu_decode(lib, hip, data, device)
        st, want, w, h, gray = orc.jpeg_decode("orc", data)
        assert rc == 0 and st > 0, (name, rc, st)
        assert (desc.width, desc.height) == (w, h) and (desc.pixelFormat == hip.PIX_FMT_MONOCHROME) == bool(gray), name
        assert np.array_equal(got, want), (name, int((got != want).sum()))
    if "prog" in extra:      # a progressive file: its scans are decoded on the host, the planes are libjpeg's (tests/test_jpeg_progressive.py)
        rc, got, desc = _gpu_decode(lib, hip, extra["prog"], device)
        st, want, w, h, gray = orc.jpeg_decode("lj", extra["prog"])
        assert rc == 0 and st > 0 and (desc.width, desc.height) == (w, h) and np.array_equal(got, want)
    if "s444" in extra:                          # 4:4:4: the reference's decompressImage fails as well (jpegdecoderhelper.cpp:283-289)
        assert _gpu_decode(lib, hip, extra["s444"], device)[0] == hip.UNKNOWN_ERROR


_PIL_RESTART = r"""
import io, sys, numpy as np
from PIL import Image
d = np.load(sys.argv[1]); out = {}
y, u, v = d["y"], d["u"], d["v"]
ycc = np.stack([y, np.repeat(np.repeat(u, 2, 0), 2, 1), np.repeat(np.repeat(v, 2, 0), 2, 1)], -1)
im, g = Image.fromarray(ycc, mode="YCbCr"), Image.fromarray(y, mode="L")
def enc(img, **kw):
    b = io.BytesIO(); img.save(b, "JPEG", **kw); return np.frombuffer(b.getvalue(), np.uint8)
out["rows_q95"] = enc(im, quality=95, subsampling=2, restart_marker_rows=1)
out["rows2_q75_opt"] = enc(im, quality=75, subsampling=2, restart_marker_rows=2, optimize=True)
out["blocks1_q90"] = enc(im, quality=90, subsampling=2, restart_marker_blocks=1)        # an interval per MCU
out["blocks7_q100"] = enc(im, quality=100, subsampling=2, restart_marker_blocks=7)
out["blocks4096_q85"] = enc(im, quality=85, subsampling=2, restart_marker_blocks=4096)  # long intervals: synchronisation inside them
out["gray_rows_q90"] = enc(g, quality=90, restart_marker_rows=1)
out["gray_blocks3_q50"] = enc(g, quality=50, restart_marker_blocks=3)
np.savez(sys.argv[2], **out)
"""


@pytest.mark.parametrize("size", [(1920, 1080), (200, 120), (16, 16)])
def test_decoder_reads_restart_intervals(hip, orc, tmp_path, size):
    """files with DRI / RSTn markers (libjpeg-turbo through Pillow writes them; cameras do): planes identical to the CPU restatement,
    which tests/test_jpeg_oracle.py holds equal to libjpeg on such files; then corrupted copies must come back with a status"""
    import subprocess, sys
    from tests.test_jpeg_oracle import _content
    lib = hip.load()
    w, h = size
    y, u, v = _content("smooth", w, h, np.random.RandomState(w))
    np.savez(tmp_path / "in.npz", y=y, u=u, v=v)
    try:
        subprocess.check_call([sys.executable, "-c", _PIL_RESTART, str(tmp_path / "in.npz"), str(tmp_path / "out.npz")], stderr=subprocess.DEVNULL)
    except (subprocess.CalledProcessError, OSError):
        pytest.skip("Pillow not usable here")
    res = np.load(tmp_path / "out.npz")
    for name in res.files:
        data = res[name].tobytes()
        assert b"\xff\xdd" in data[:2000]
        st, want, dw, dh, gray = orc.jpeg_decode("orc", data)
        assert st > 0 and (dw, dh) == (w, h), name
        for device in (True, False):
            rc, got, desc = _gpu_decode(lib, hip, data, device)
            assert rc == 0 and (desc.width, desc.height) == (w, h), (name, rc)
            assert np.array_equal(got, want), (name, int((got != want).sum()))
    # damage: a marker removed, a marker added, a truncated file, bytes flipped inside an interval -- a status, never a fault
    data = bytearray(res["rows_q95"].tobytes())
    sos = data.find(b"\xff\xda")
    first_rst = data.find(b"\xff\xd0", sos)
    rng = np.random.RandomState(5)
    variants = [bytes(data[:first_rst] + data[first_rst + 2:]), bytes(data[:first_rst] + b"\xff\xd3" + data[first_rst:]), bytes(data[:len(data) * 2 // 3])]
    for _ in range(12):
        d = bytearray(data)
        for _ in range(4):
            d[rng.randint(sos + 14, len(d) - 2)] = rng.randint(0, 256)
        variants.append(bytes(d))
    for k, bad in enumerate(variants):
        rc, got, desc = _gpu_decode(lib, hip, bad, True)
        assert rc in (0, hip.UNKNOWN_ERROR, hip.ERROR_UNSUPPORTED_FEATURE), (k, rc)


def test_decoder_rejects_malformed_input(hip, orc):
    lib = hip.load()
    rng = np.random.RandomState(4)
    y, u, v = _content("smooth", 64, 48, rng)
    uv = np.ascontiguousarray(np.concatenate([u.reshape(-1), v.reshape(-1)]))
    good = orc.jpeg_encode("orc", np.ascontiguousarray(y), uv, 64, 48, 90)
    desc = hip.Image()
    for bad in (b"", b"\xff\xd8", b"notajpegnotajpeg", good[:300], good[:len(good) // 2]):
        buf = np.frombuffer(bad + b"\0" * 8, np.uint8)
        rc = lib.uhdr_hip_jpeg_decode(C.c_void_p(buf.ctypes.data), len(bad), None, 0, C.byref(desc), hip.MEM_HOST, None)
        assert rc in (hip.UNKNOWN_ERROR, hip.ERROR_BAD_PTR), (len(bad), rc)
    # entropy-coded data cut short but with a valid EOI glued on: the scan ends before the last block
    cut = good[:len(good) - 200] + b"\xff\xd9"
    rc, _, _ = _gpu_decode(lib, hip, cut, True)
    assert rc == hip.UNKNOWN_ERROR
    assert lib.uhdr_hip_jpeg_decode(None, 10, None, 0, C.byref(desc), hip.MEM_HOST, None) == hip.ERROR_BAD_PTR


def test_4k_round_trip_through_the_device_codec(hip, orc):
    """encode a 4K frame on the device, decode it on the device, compare with the oracle's decode of the same bytes; then
    the decoded planes feed applyGainMap (the decode path of jpegr.cpp:796-801)"""
    from tests.gpu_util import dev_empty, gpu_apply, stream_ptr, to_dev, to_host
    from tests.test_gpu_parity import smooth_frame
    lib = hip.load()
    w, h = 3840, 2160
    _, yuv = smooth_frame(w, h, 8)
    rc, n, data = _gpu_encode(lib, hip, yuv[:w * h], yuv[w * h:], w, h, 95, w, w // 2, True)
    assert rc == 0
    rc, got, desc = _gpu_decode(lib, hip, data, True)
    st, want, dw, dh, gray = orc.jpeg_decode("orc", data)
    assert rc == 0 and st == w * h * 3 // 2 and np.array_equal(got, want)
    assert np.abs(got[:w * h].astype(int) - yuv[:w * h].astype(int)).mean() < 2.0


def test_decoder_survives_corrupted_files(hip, orc):
    """450 single- and multi-byte corruptions of valid files (header and entropy-coded data alike): every call must come
    back with a status of the documented set -- no fault, no hang -- and an untouched file still decodes afterwards.
    (What a corrupt file decodes TO is not compared: libjpeg and this decoder resynchronise differently after an error.)"""
    lib = hip.load()
    rng = np.random.RandomState(77)
    allowed = {0, hip.UNKNOWN_ERROR, hip.ERROR_UNSUPPORTED_FEATURE, hip.ERROR_RESOLUTION_MISMATCH, hip.ERROR_BAD_PTR}
    seeds = []
    for kind, (w, h), gray in (("smooth", (64, 48), False), ("noise", (40, 24), True), ("extreme", (130, 66), False)):
        y, u, v = _content(kind, w, h, rng)
        uv = None if gray else np.ascontiguousarray(np.concatenate([u.reshape(-1), v.reshape(-1)]))
        seeds.append(orc.jpeg_encode("orc", np.ascontiguousarray(y), uv, w, h, 80))
    outcomes = {}
    for good in seeds:
        for _ in range(150):
            b = bytearray(good)
            for _k in range(rng.randint(1, 4)):
                pos = rng.randint(2, len(b))
                b[pos] = rng.randint(0, 256) if rng.rand() < 0.7 else b[pos] ^ (1 << rng.randint(0, 8))
            buf = np.frombuffer(bytes(b) + b"\0" * 8, np.uint8)
            desc = hip.Image()
            rc = lib.uhdr_hip_jpeg_decode(C.c_void_p(buf.ctypes.data), len(b), None, 0, C.byref(desc), hip.MEM_HOST, None)
            if rc == hip.ERROR_INSUFFICIENT_RESOURCE:
                need = desc.width * desc.height * 2 + 64
                if need > (1 << 23):
                    continue          # a corrupted size field asking for a huge image: the probe already answered
                out = np.zeros(need, np.uint8)
                rc = lib.uhdr_hip_jpeg_decode(C.c_void_p(buf.ctypes.data), len(b), C.c_void_p(out.ctypes.data), need, C.byref(desc), hip.MEM_HOST, None)
            assert rc in allowed, rc
            outcomes[rc] = outcomes.get(rc, 0) + 1
        rc, got, _ = _gpu_decode(lib, hip, good, False)
        assert rc == 0 and np.array_equal(got, orc.jpeg_decode("orc", good)[1])
    print("corrupted-file outcomes by status:", outcomes)
    assert outcomes.get(0, 0) > 0 and outcomes.get(hip.UNKNOWN_ERROR, 0) > 0


def test_random_files_decode_like_libjpeg(hip, orc, tmp_path):
    """400 random files (oracle encoder and Pillow / libjpeg-turbo: standard and optimised tables, restart intervals from one MCU to
    thousands, 2x2 to 2400x1400, qualities 1-100, noise / smooth / flat content, colour and single-plane): the device decoder's
    planes equal libjpeg's in device and in host memory.  `python tests/stress_jpeg_dec.py 20000 11` is the long form of this test
    (40 000 identical decodes on an MI355X after the decoder's synchronisation passes were rewritten)."""
    try:
        import PIL  # noqa: F401
    except ImportError:
        pytest.skip("Pillow not usable here")
    from tests import stress_jpeg_dec
    decoded, mismatches = stress_jpeg_dec.run(400, 3, dump_dir=str(tmp_path), damage=True)   # (each file also damaged: a status, never a fault)
    assert mismatches == 0 and decoded == 800


def test_random_images_encode_like_libjpeg(hip, orc):
    """250 random images (2x2 to 1500x900, tight / aligned / padded / odd strides, qualities 1-100, noise / smooth / extreme / flat
    content, colour and single-plane, device and host memory): the device encoder's bytes equal the CPU restatement's (which
    tests/test_jpeg_oracle.py pins to libjpeg).  Written when the bit counts moved into the DCT kernel and the two prefix sums into
    the kernels that consume them."""
    lib = hip.load()
    import os
    rng = np.random.RandomState(int(os.environ.get("UHDR_ENC_SWEEP_SEED", "21")))
    kinds = ["smooth", "noise", "extreme", "flat"]
    for it in range(int(os.environ.get("UHDR_ENC_SWEEP", "250"))):   # (UHDR_ENC_SWEEP=8000 is the long form: no mismatch)
        big = rng.rand() < 0.06
        w = 2 * rng.randint(1, 750 if big else 100)
        h = 2 * rng.randint(1, 450 if big else 80)
        kind = kinds[rng.randint(0, 4)]
        y, u, v = _content(kind, w, h, rng)
        aw, acw = (w + 15) // 16 * 16, (w // 2 + 7) // 8 * 8
        ls, cs = [(w, w // 2), (aw, acw), (aw + 16, acw + 8), (w + 2, w // 2 + 1)][rng.randint(0, 4)]
        yb, ub = _planes(y, u, v, ls, cs, rng)
        q = int(rng.choice([rng.randint(1, 101), 95, 100, 85, 50]))
        device = bool(rng.randint(0, 2))
        if rng.rand() < 0.25:
            rc, n, got = _gpu_encode(lib, hip, yb, None, w, h, q, ls, 0, device)
            want = orc.jpeg_encode("orc", yb, None, w, h, q, ls)
        else:
            rc, n, got = _gpu_encode(lib, hip, yb, ub, w, h, q, ls, cs, device)
            want = orc.jpeg_encode("orc", yb, ub, w, h, q, ls, cs)
        assert rc == 0 and n == len(want) and got == want, (it, kind, w, h, ls, cs, q, device, n, len(want))
