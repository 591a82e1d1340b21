"""-m gpu: the C++ host shim (ultrahdr::UltraHdrHip, include/ultrahdr_hip/ultrahdr_hip.h) driven by a C++ program
shaped like the reference's own harness, and the HBM-side synthetic frame generator used by bench.py."""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def test_cpp_shim_on_reference_fixture(hip, orc, tmp_path):
    exe = str(tmp_path / "shim_test")
    pkg = os.path.join(ROOT, "libultrahdr_dev_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "shim_test.cpp"), "-o", exe,
                           "-L" + pkg, "-lultrahdr_shim", "-luhdr_hip", "-Wl,-rpath," + pkg])
    g = os.path.join(ROOT, "tests", "golden")
    env = dict(os.environ)
    # a pure C++ host links the system ROCm runtime; nothing of torch is involved in this process
    r = subprocess.run([exe, os.path.join(g, "raw_p010_image.p010"), os.path.join(g, "raw_yuv420_image.yuv420"), str(tmp_path)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    # golden md5s of the real reference
    assert md5(tmp_path / "map_hlg.bin") == "32e38116ea48d76872b525663d33d137"
    assert md5(tmp_path / "apply_hlg_exact.bin") == "dfd56dc878636c93b7a92b45cf35ee53"
    assert md5(tmp_path / "apply_f16_exact.bin") == "237aa8455a4d5b18136f36e2c59eb8c3"
    assert md5(tmp_path / "tonemap.bin") == "f8a112ef5d54c8df357fd082b22c5397"
    # FAST PQ within 1 LSB of the reference bytes (re-created by the oracle, itself md5-pinned)
    w, h = 1280, 720
    yuv = np.fromfile(os.path.join(g, "raw_yuv420_image.yuv420"), np.uint8)
    gmap = np.fromfile(tmp_path / "map_hlg.bin", np.uint8).reshape(h // 4, w // 4)
    mb = float(np.float32(1000.0) / np.float32(203.0))
    omd = orc.Metadata(mb, 1.0, 1.0, 0.0, 0.0, 1.0, mb, 1)
    st, ref, _ = orc.apply("orc_", orc.yuv420_image(yuv, w, h, 0), gmap, omd, orc.OUT_HDR_PQ, 3.4028234663852886e38, threads=8)
    assert hashlib.md5(ref.tobytes()).hexdigest() == "87ffff4015d36d3fd664bc918039c00d"
    from tests.gpu_util import diff_1010102
    worst, frac, alpha_ok = diff_1010102(np.fromfile(tmp_path / "apply_pq_fast.bin", np.uint32), ref.view(np.uint32))
    assert alpha_ok and worst <= 1
    # convertYuv against the oracle
    cv = yuv.copy()
    img = orc.yuv420_image(cv, w, h, 0)
    assert orc.load().orc_convertYuv(C.byref(img), 0, 1) == 0
    assert np.array_equal(np.fromfile(tmp_path / "convert_709_601.bin", np.uint8), cv)
    # editorhelper free functions against the oracle (itself checked against the reference's editorhelper.cpp)
    L = orc.load()
    src = orc.Image(yuv.ctypes.data, w, h, 0, yuv.ctypes.data + w * h, w, w // 2, orc.FMT_YUV420)
    for fname, fn, args, nbytes in (("rotate90.bin", L.orc_rotate, (90,), w * h * 3 // 2), ("crop.bin", L.orc_crop, (100, 739, 40, 519), 640 * 480 * 3 // 2),
                                    ("mirror_h.bin", L.orc_mirror, (1,), w * h * 3 // 2), ("resize.bin", L.orc_resize, (640, 360), 640 * 360 * 3 // 2)):
        o = np.zeros(w * h * 3 // 2 + 64, np.uint8)
        oi = orc.Image(o.ctypes.data, 0, 0, -1, None, 0, 0, -1)
        assert fn(C.byref(src), *args, C.byref(oi)) == 0
        assert np.array_equal(np.fromfile(tmp_path / fname, np.uint8), o[:nbytes]), fname
    arr = (orc.Effect * 4)(orc.Effect(3, w * 3 // 4, h * 3 // 4, 0, 0), orc.Effect(1, 0, 0, 0, 0), orc.Effect(2, 90, 0, 0, 0), orc.Effect(0, 20, 149, 10, 99))
    o = np.zeros(w * h * 3 // 2 + 64, np.uint8)
    oi = orc.Image(o.ctypes.data, 0, 0, -1, None, 0, 0, -1)
    assert L.orc_add_effects(C.byref(src), arr, 4, C.byref(oi)) == 0 and (oi.width, oi.height) == (130, 90)
    assert np.array_equal(np.fromfile(tmp_path / "effects_chain.bin", np.uint8), o[:130 * 90 * 3 // 2])
    # JPEG helper mirrors against the CPU checker (pinned to libjpeg by tests/test_jpeg_oracle.py)
    want = orc.jpeg_encode("orc", np.ascontiguousarray(gmap.reshape(-1)), None, w // 4, h // 4, 85)
    assert open(tmp_path / "map_q85.jpg", "rb").read() == want
    want = orc.jpeg_encode("orc", yuv[:w * h], yuv[w * h:], w, h, 95, icc=b"icc!\0")
    assert open(tmp_path / "sdr_q95.jpg", "rb").read() == want
    st, planes, dw, dh, gray = orc.jpeg_decode("orc", want)
    assert st > 0 and np.array_equal(np.fromfile(tmp_path / "sdr_q95_decoded.bin", np.uint8), planes)
    assert np.array_equal(np.fromfile(tmp_path / "sdr_q95_rgba.bin", np.uint8).reshape(h, w, 4), orc.ycc420_to_rgba(planes, w, h))
    st, planes, dw, dh, gray = orc.jpeg_decode("orc", open(tmp_path / "map_q85.jpg", "rb").read())
    assert st > 0 and gray and np.array_equal(np.fromfile(tmp_path / "map_q85_decoded.bin", np.uint8), planes)
    # JpegRHip: every encodeJPEGR overload and decodeJPEGR against the CPU restatement (oracle/jpegr_oracle.py, pinned to the reference's
    # sample file by tests/test_jpegr_container.py), on the reference's own 1280x720 fixture pair
    from oracle import jpegr_oracle as J
    p010 = np.fromfile(os.path.join(g, "raw_p010_image.p010"), np.uint16)
    rd = lambda n: open(tmp_path / n, "rb").read()
    exif = b"Exif\0\0MM\0*\0\0\0\x08\0\0"
    assert rd("api0.jpgr") == J.encode_api0(p010, w, h, 2, 1, 90)
    api1 = J.encode_api1(p010, yuv, w, h, 0, 2, 1, 90, exif=exif)
    assert rd("api1.jpgr") == api1
    st, ref, ow, oh, gamut, md = J.decode(api1, orc.OUT_HDR_HLG, 3.4028234663852886e38)
    assert st == 0 and np.array_equal(np.fromfile(tmp_path / "api1_decoded_hlg.bin", np.uint8), ref)
    st, sref, _, _, _, _ = J.decode(api1, orc.OUT_SDR, 3.4028234663852886e38)
    assert st == 0 and np.array_equal(np.fromfile(tmp_path / "api1_decoded_sdr.bin", np.uint8), sref)
    gj = api1[J.find_images(api1)[1][0]:]
    assert np.array_equal(np.fromfile(tmp_path / "api1_decoded_map.bin", np.uint8), orc.jpeg_decode("orc", gj)[1])
    plain = orc.jpeg_encode("orc", yuv[:w * h], yuv[w * h:], w, h, 95)
    assert rd("api2.jpgr") == J.encode_api2(p010, yuv, w, h, 0, 2, plain, 0, 1)
    assert rd("api3.jpgr") == J.encode_api3(p010, w, h, 2, plain, 0, 1)
    i1 = J.info(api1)
    pj, gjj = api1[:i1[0]["size"]], api1[i1[1]["offset"]:]
    assert rd("api4.jpgr") == J.encode_api4(pj, 0, gjj, md) and rd("api4.jpgr")[:len(api1)] != api1     # EXIF lifted, XMP segments now nested: a different file
    mdx = dict(version="1.0", max=np.float32(mb), min=np.float32(1.0), gamma=np.float32(1.0), off_sdr=np.float32(0), off_hdr=np.float32(0),
               capmin=np.float32(1.0), capmax=np.float32(mb))
    assert rd("apix.jpgr") == J.encode_apix(yuv, w, h, 0, gmap, mdx, 90)


def test_hbm_synthetic_frames_match_the_survey_lcg(hip, orc):
    from libultrahdr_dev_amd import synth
    for (w, h, seed) in ((640, 480, 1234), (3840, 2160, 1235)):
        p, y = synth.lcg_frame(w, h, seed)
        op, oy = orc.lcg_frame(w, h, seed)
        assert np.array_equal(p.cpu().numpy().view(np.uint16), op) and np.array_equal(y.cpu().numpy(), oy)
