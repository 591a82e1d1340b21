"""Progressive JPEG input (SOF2; csrc/uhdr_jpeg_prog.cpp): the reference decodes whatever libjpeg reads
(lib/src/jpegdecoderhelper.cpp:190-320), so a JPEG/R file with a progressive primary image must decode.

CPU (no GPU): the host-side entropy decoder's coefficients against the image's libjpeg (jpeg_read_coefficients behind
oracle/jpeg_libjpeg_harness.c) -- every scan type of the standard progression (interleaved DC first, DC refinement, AC first and AC
refinement bands per component), colour and grayscale, MCU-aligned and ragged sizes, optimised tables, restart intervals.
The files are written by Pillow's libjpeg-turbo (progressive=True): no progressive fixture exists in the reference's tests
(tests/data holds baseline files only), so parity is pinned on libjpeg itself, like the baseline corpus.
GPU: the decoded planes against libjpeg's raw-data output, through uhdr_hip_jpeg_decode and inside a JPEG/R file."""
import ctypes as C
import io

import numpy as np
import pytest

PIL = pytest.importorskip("PIL")
from PIL import Image


def _image(w, h, seed, gray):
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = (96 + 80 * np.sin(xx / 9.0 + seed) * np.cos(yy / 13.0) + 14 * rng.randn(h, w)).clip(0, 255)
    if gray:
        return Image.fromarray(base.astype(np.uint8), mode="L")
    rgb = np.stack([base, (base * 0.7 + 40 * np.cos(xx / 17.0)).clip(0, 255), (255 - base * 0.8 + 9 * rng.randn(h, w)).clip(0, 255)], axis=-1)
    return Image.fromarray(rgb.astype(np.uint8), mode="RGB")


def _progressive(w, h, seed, gray=False, quality=90, optimize=False, restart=0):
    b = io.BytesIO()
    kw = dict(quality=quality, progressive=True, optimize=optimize)
    if not gray:
        kw["subsampling"] = "4:2:0"
    if restart:
        kw["restart_marker_blocks"] = restart
    _image(w, h, seed, gray).save(b, "JPEG", **kw)
    data = b.getvalue()
    assert b"\xff\xc2" in data[:600]   # SOF2
    return data


CASES = [(64, 48, False, 90, False, 0), (70, 34, False, 75, True, 0), (16, 16, False, 95, False, 0), (129, 97, False, 60, False, 0),
         (64, 48, True, 90, False, 0), (33, 21, True, 80, True, 0), (320, 240, False, 92, False, 0), (200, 120, False, 85, False, 3),
         (96, 80, True, 70, False, 2), (8, 8, False, 50, False, 0), (1920, 1080, False, 95, True, 0)]


@pytest.mark.parametrize("case", CASES)
def test_progressive_coefficients_equal_libjpegs(orc, case):
    from libultrahdr_dev_amd import api
    lib = api.load()
    lj = orc.load_libjpeg()
    if lj is None:
        pytest.skip("the image's libjpeg harness is not built")
    w, h, gray, q, opt, rst = case
    data = _progressive(w, h, 11 + w, gray, q, opt, rst)
    buf = np.frombuffer(data, np.uint8)
    nblk = ((w + 7) // 8) * ((h + 7) // 8) if gray else ((w + 15) // 16) * ((h + 15) // 16) * 6
    want = np.zeros(nblk * 64, np.int16)
    pw, ph, pg = C.c_int(), C.c_int(), C.c_int()
    lj.lj_jpeg_coefficients.restype = C.c_long
    lj.lj_jpeg_coefficients.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    assert lj.lj_jpeg_coefficients(buf.ctypes.data, buf.size, want.ctypes.data, nblk, C.byref(pw), C.byref(ph), C.byref(pg)) == nblk
    got = np.zeros(nblk * 64, np.int16)
    nb, gw, gh, gg = C.c_size_t(), C.c_int(), C.c_int(), C.c_int()
    rc = lib.uhdr_hip_jpeg_progressive_coefficients(C.c_void_p(buf.ctypes.data), buf.size, C.c_void_p(got.ctypes.data), nblk, C.byref(nb),
                                                    C.byref(gw), C.byref(gh), C.byref(gg))
    assert rc == 0 and nb.value == nblk and (gw.value, gh.value, gg.value) == (w, h, int(gray)) == (pw.value, ph.value, pg.value)
    # blocks beyond a component's own extent (the MCU padding an interleaved scan carries) exist in the scans' DC prediction but
    # not in libjpeg's coefficient arrays, and nothing of them reaches the image: left out of the comparison
    real = np.ones(nblk, bool)
    if not gray:
        mx = (w + 15) // 16
        for b in range(nblk):
            mcu, k = divmod(b, 6)
            mr, mc = divmod(mcu, mx)
            if k < 4:
                real[b] = (2 * mr + (k >> 1)) < (h + 7) // 8 and (2 * mc + (k & 1)) < (w + 7) // 8
            else:
                real[b] = mr < ((h + 1) // 2 + 7) // 8 and mc < ((w + 1) // 2 + 7) // 8
    got = (got.reshape(nblk, 64) * real[:, None]).reshape(-1)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "%d coefficients differ, first at block %d position %d: %d vs libjpeg %d" % (
        bad.size, bad[0] // 64, bad[0] % 64, got[bad[0]], want[bad[0]])


def test_progressive_entry_point_statuses(orc):
    from libultrahdr_dev_amd import api
    lib = api.load()
    nb, gw, gh, gg = C.c_size_t(), C.c_int(), C.c_int(), C.c_int()
    call = lambda d, cap=0, out=None: lib.uhdr_hip_jpeg_progressive_coefficients(
        C.c_void_p(np.frombuffer(d, np.uint8).ctypes.data), len(d), out, cap, C.byref(nb), C.byref(gw), C.byref(gh), C.byref(gg))
    data = _progressive(64, 48, 3)
    assert call(data) == api.ERROR_INSUFFICIENT_RESOURCE and nb.value == 4 * 3 * 6     # the count is reported, the buffer was too small
    b = io.BytesIO()
    _image(64, 48, 3, False).save(b, "JPEG", quality=90, subsampling="4:2:0")
    assert call(b.getvalue()) == api.ERROR_UNSUPPORTED_FEATURE                          # baseline: decoded on the device
    assert call(data[:len(data) // 2]) == api.UNKNOWN_ERROR                             # truncated: a status, never a fault
    b = io.BytesIO()
    _image(64, 48, 3, False).save(b, "JPEG", quality=90, progressive=True, subsampling="4:4:4")
    assert call(b.getvalue()) == api.UNKNOWN_ERROR                                      # the reference refuses the sampling too


@pytest.mark.gpu
@pytest.mark.parametrize("device", [True, False])
def test_gpu_decodes_progressive_files_like_libjpeg(hip, orc, device):
    import torch
    lib = hip.load()
    for (w, h, gray, q, opt, rst) in CASES:
        data = _progressive(w, h, 11 + w, gray, q, opt, rst)
        st, want, ww, wh, wg = orc.jpeg_decode("lj", data)
        assert st > 0 and (ww, wh, wg) == (w, h, int(gray))
        need = w * h if gray else w * h + 2 * ((w * h) // 4)     # (uhdr_hip_jpeg_decode's own size rule)
        buf = np.frombuffer(data, np.uint8)
        desc = hip.Image()
        if device:
            out = torch.full((need + 64,), 0xCD, dtype=torch.uint8, device="cuda")
            rc = lib.uhdr_hip_jpeg_decode(C.c_void_p(buf.ctypes.data), buf.size, C.c_void_p(out.data_ptr()), need, C.byref(desc), hip.MEM_DEVICE, None)
            torch.cuda.synchronize()
            got = out.cpu().numpy()[:need]
        else:
            got = np.full(need, 0xCD, np.uint8)
            rc = lib.uhdr_hip_jpeg_decode(C.c_void_p(buf.ctypes.data), buf.size, C.c_void_p(got.ctypes.data), need, C.byref(desc), hip.MEM_HOST, None)
        assert rc == 0 and (desc.width, desc.height) == (w, h)
        if not gray and (w % 2 or h % 2):
            continue   # (odd sizes: the reference's own chroma copy is what differs, not the decoder; covered for baseline files)
        assert np.array_equal(got, want[:need]), (w, h, gray, int((got != want[:need]).sum()))


@pytest.mark.gpu
def test_gpu_decodes_a_jpegr_file_with_a_progressive_primary_image(hip, orc):
    """decodeJPEGR on a file whose primary image an editor re-saved progressively: the rendition of the restatement (libjpeg
    planes -> applyGainMap), bit for bit in EXACT mode"""
    from oracle import jpegr_oracle as J
    from tests.test_jpegr_container import SAMPLE_MD, _gpu_decode
    lib = hip.load()
    w, h = 256, 128
    primary = _progressive(w, h, 5, gray=False, quality=92)
    rng = np.random.RandomState(2)
    gray = orc.jpeg_encode("orc", rng.randint(0, 256, (w // 4) * (h // 4)).astype(np.uint8), None, w // 4, h // 4, 85)
    data = J.append_gainmap(primary, gray, SAMPLE_MD)
    for fmt in (hip.OUTPUT_HDR_HLG, hip.OUTPUT_HDR_PQ, hip.OUTPUT_HDR_LINEAR):
        st, want, ww, wh, gamut, omd = J.decode(data, fmt, 3.4028234663852886e38)
        rc, got, dest, md = _gpu_decode(lib, hip, data, fmt, 3.4028234663852886e38, hip.APPLY_EXACT, True)
        assert rc == st == 0 and (dest.width, dest.height) == (w, h)
        assert np.array_equal(got, want), (fmt, int((got != want).sum()))
