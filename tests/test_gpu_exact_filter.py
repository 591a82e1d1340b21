"""-m gpu: EXACT apply behind its f32 pre-filter (csrc/uhdr_kernels.hip: k_apply_px_est / k_apply_resolve).

The EXACT mode first evaluates every pixel on the f32 units and sends only the pixels whose
integer code (half-precision pattern) that estimate cannot settle to the double-precision path.  Bit-exactness then rests on

 (1) the error bounds the doubt test assumes (kEstRel, kEstOetfAbs / kEstOetfRel in the kernel source) -- measured here for EVERY float
     of each function's domain against the exact device functions (themselves pinned to glibc in test_gpu_transfer_exhaustive.py);
 (2) filtered == unfiltered on whole images: random frames at several scales and display boosts, a batch, and frames built so that
     every pixel is in doubt (the lists overflow and the resolve kernel sweeps the image).
The reference-md5 and oracle comparisons of EXACT mode in test_gpu_parity.py / test_jpegr_container.py run through the same path.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FLT_MAX = 3.4028234663852886e38
# the budget of csrc/uhdr_kernels.hip, restated: the test fails if the measured errors do not fit into it
EST_REL = 2.0e-6
EST_OETF_ABS, EST_OETF_REL = 8.0e-4, 6.0e-7


def _eval(lib, fn, x):
    out = torch.empty_like(x)
    rc = lib.uhdr_hip_eval_transfer(fn, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), x.numel(), 1.0, 4.0,
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    return out


def _all_floats(lo_bits, hi_bits, chunk=1 << 26):
    for b in range(lo_bits, hi_bits + 1, chunk):
        n = min(chunk, hi_bits + 1 - b)
        yield (torch.arange(n, dtype=torch.int32, device="cuda") + b).view(torch.float32)


def _worst_rel(lib, fast, exact, ranges):
    worst = 0.0
    for lo, hi in ranges:
        for x in _all_floats(lo, hi):
            a, b = _eval(lib, fast, x).double(), _eval(lib, exact, x).double()
            m = b != 0
            worst = max(worst, float(((a - b).abs()[m] / b[m].abs()).max().item()))
            assert bool((a[~m] == 0).all())
    return worst


def test_estimate_error_budget(hip):
    lib = hip.load()
    # sRGB EOTF: every float from 2^-10 (below 1/255 the input is 0) to 1
    e_srgb = _worst_rel(lib, 20, 10, [(int(np.float32(2.0 ** -10).view(np.uint32)), 0x3F800000)])
    # the gain factor 2^x: every float with 2^-20 <= |x| <= 32 (an argument below 2^-20 gives 1 to 1e-6), both signs
    lo = int(np.float32(2.0 ** -20).view(np.uint32))
    e_exp = _worst_rel(lib, 26, 16, [(lo, 0x42000000), (lo + 0x80000000 - (1 << 32), 0xC2000000 - (1 << 32))])
    z = torch.tensor([0.0, 1e-7, -1e-7, 5e-7], dtype=torch.float32, device="cuda")
    e_exp = max(e_exp, float(((_eval(lib, 26, z).double() - _eval(lib, 16, z).double()).abs()).max().item()))
    # lin = (srgb * factor) / boost: the product and the quotient are the reference's own operations on perturbed operands, each
    # rounding of a perturbed value adds at most 2^-24; the cell kernel multiplies the factor by the rounded reciprocal of the boost
    # instead (three roundings more)
    # the packed cell (est_cell_pk) forms the exponent's argument with a two-float product whose rounded result can sit one ulp from
    # the reference's (once in 2^24 pixels); it is used for arguments up to 6, where one ulp moves the factor by ln 2 * 6 * 2^-23
    e_arg = 0.6931471805599453 * 6.0 * 2.0 ** -23
    total = e_srgb + e_exp + 5 * 2.0 ** -24 + e_arg
    print("estimate: sRGB EOTF %.3g, 2^x %.3g, argument %.3g relative; lin within %.3g (budget %.3g)" % (e_srgb, e_exp, e_arg, total, EST_REL))
    assert total * 1.25 <= EST_REL, (e_srgb, e_exp)

    # hlgOetf: |fast - exact| in code values against the part of the budget that is not the input's error (1023 / 4 * EST_REL)
    worst_margin = 0.0
    for x in _all_floats(0, 0x42800000):        # every float in [0, 64]
        a, b = _eval(lib, 24, x).double() * 1023.0, _eval(lib, 14, x).double() * 1023.0
        # + one rounding of e * 1023 on each side
        err = (a - b).abs() + 2.0 ** -24 * b
        # sensitivity x h'(x) of the code value to the input's relative error: h/2 below the junction, a 12x / (12x - b) above
        xd = x.double()
        sens = torch.where(xd <= 1.0 / 12.0, 0.5 * b, 1023.0 * 0.17883277 * 12.0 * xd / (12.0 * xd - 0.28466892).clamp_min(0.7))
        need = err + sens * EST_REL
        have = EST_OETF_ABS + a * EST_OETF_REL
        worst_margin = max(worst_margin, float((need / have).max().item()))
    print("hlgOetf estimate: needs at most %.2f of the doubt interval" % worst_margin)
    assert worst_margin <= 0.9, worst_margin

    # pqOetf: pq_oetf_est against the exact function, plus the input's error through S(x) = d ln(code) / d ln(x)
    m1, m2 = 2610.0 / 16384.0, 2523.0 / 4096.0 * 128.0
    c1, c2, c3 = 3424.0 / 4096.0, 2413.0 / 4096.0 * 32.0, 2392.0 / 4096.0 * 32.0
    worst_margin = worst_err = 0.0
    for x in _all_floats(0, 0x42800000):
        a, b = _eval(lib, 27, x).double() * 1023.0, _eval(lib, 15, x).double() * 1023.0
        err = (a - b).abs() + 2.0 ** -24 * b
        p = x.double().clamp_min(1e-300).pow(m1)
        sens = b * m2 * m1 * p * (c2 / (c1 + c2 * p) - c3 / (1.0 + c3 * p))
        need = err + sens * EST_REL
        have = EST_OETF_ABS + a * EST_OETF_REL
        worst_margin = max(worst_margin, float((need / have).max().item()))
        worst_err = max(worst_err, float((a - b).abs().max().item()))
    print("pqOetf estimate: within %.3g codes of the exact function; needs at most %.2f of the doubt interval" % (worst_err, worst_margin))
    assert worst_margin <= 0.9, worst_margin


def _apply(lib, hip, yuv, w, h, gmap, md, fmt, boost, mode):
    from tests.gpu_util import gpu_apply, to_dev
    dy, dmap = to_dev(yuv), to_dev(gmap)
    yi = hip.yuv420_image(dy.data_ptr(), w, h, hip.CG_BT709)
    st, out, _ = gpu_apply(lib, yi, dmap, gmap.shape[1], gmap.shape[0], md, fmt, boost, mode)
    assert st == 0
    return out


@pytest.mark.parametrize("fmt", [1, 2, 3, 4])
@pytest.mark.parametrize("scale,boost", [(4, FLT_MAX), (4, 2.0), (1, FLT_MAX), (3, 1.0), (8, FLT_MAX)])
def test_filtered_equals_unfiltered(hip, orc, fmt, scale, boost):
    lib = hip.load()
    mw, mh = 160, 90
    w, h = mw * scale, mh * scale
    if w % 2 or h % 2:
        w, h, mw, mh = w * 2, h * 2, mw * 2, mh * 2
    _, yuv = orc.lcg_frame(w, h, 5 + scale)
    gmap = np.random.RandomState(scale * 10 + fmt).randint(0, 256, (mh, mw)).astype(np.uint8)
    md = hip.metadata(np.float32(1000.0 / 203.0), np.float32(0.8))
    a = _apply(lib, hip, yuv, w, h, gmap, md, fmt, boost, hip.APPLY_EXACT)
    b = _apply(lib, hip, yuv, w, h, gmap, md, fmt, boost, hip.APPLY_EXACT_UNFILTERED)
    assert np.array_equal(a, b), "%d bytes differ" % int((a != b).sum())
    # and again on the same stream: the first launch left the list headers cleared
    a2 = _apply(lib, hip, yuv, w, h, gmap, md, fmt, boost, hip.APPLY_EXACT)
    assert np.array_equal(a2, b)


@pytest.mark.parametrize("fmt", [2, 3, 4])
def test_every_pixel_in_doubt_overflows_into_a_sweep(hip, orc, fmt):
    """white with gain 1 and display boost == content boost is linear 1.0: code value 1023.0 to within the estimate's error, in
    doubt for every pixel -- far more than the lists hold"""
    from tests.test_gpu_parity import _oracle_apply
    lib = hip.load()
    w, h = 512, 256
    yuv = np.concatenate([np.full(w * h, 255, np.uint8), np.full(w * h // 2, 128, np.uint8)])
    gmap = np.full((h // 4, w // 4), 255, np.uint8)
    maxb = np.float32(4.0)
    md = hip.metadata(maxb)
    ref = _oracle_apply(orc, yuv, w, h, gmap, maxb, fmt, FLT_MAX)
    a = _apply(lib, hip, yuv, w, h, gmap, md, fmt, FLT_MAX, hip.APPLY_EXACT)
    assert np.array_equal(a, ref)
    # half of the image in doubt, then an ordinary frame: the overflow left nothing behind
    yuv2 = yuv.copy()
    yuv2[: w * h // 2] = 200
    ref2 = _oracle_apply(orc, yuv2, w, h, gmap, maxb, fmt, FLT_MAX)
    assert np.array_equal(_apply(lib, hip, yuv2, w, h, gmap, md, fmt, FLT_MAX, hip.APPLY_EXACT), ref2)
    _, yuv3 = orc.lcg_frame(w, h, 77)
    ref3 = _oracle_apply(orc, yuv3, w, h, gmap, maxb, fmt, FLT_MAX)
    assert np.array_equal(_apply(lib, hip, yuv3, w, h, gmap, md, fmt, FLT_MAX, hip.APPLY_EXACT), ref3)


def test_boosts_beyond_the_measured_range_take_the_exact_path(hip, orc):
    """|log2 boost| > 32 is outside what the estimate's bounds were measured for: the call runs unfiltered (and still equals the oracle)"""
    from tests.test_gpu_parity import _oracle_apply
    lib = hip.load()
    w, h = 64, 32
    _, yuv = orc.lcg_frame(w, h, 3)
    gmap = np.random.RandomState(4).randint(0, 256, (h // 4, w // 4)).astype(np.uint8)
    minb, maxb = np.float32(2.0 ** -40), np.float32(2.0 ** 36)
    md = hip.metadata(maxb, minb)
    ref = _oracle_apply(orc, yuv, w, h, gmap, maxb, hip.OUTPUT_HDR_HLG, 8.0, minb)
    assert np.array_equal(_apply(lib, hip, yuv, w, h, gmap, md, hip.OUTPUT_HDR_HLG, 8.0, hip.APPLY_EXACT), ref)


def test_batch_of_frames_and_a_second_stream(hip, orc):
    from tests.gpu_util import to_dev, to_host
    lib = hip.load()
    n, w, h = 5, 384, 216
    md = hip.metadata(np.float32(10000.0 / 203.0))
    frames = [orc.lcg_frame(w, h, 100 + i)[1] for i in range(n)]
    maps = [np.random.RandomState(i).randint(0, 256, (h // 4, w // 4)).astype(np.uint8) for i in range(n)]
    dys, dms = [to_dev(f) for f in frames], [to_dev(m) for m in maps]
    ya = hip.image_array([hip.yuv420_image(d.data_ptr(), w, h, hip.CG_BT709) for d in dys])
    ma = hip.image_array([hip.mono_image(d.data_ptr(), w // 4, h // 4) for d in dms])
    res = {}
    side = torch.cuda.Stream()
    for mode, stream in ((hip.APPLY_EXACT, side), (hip.APPLY_EXACT_UNFILTERED, torch.cuda.current_stream())):
        outs = [torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda") for _ in range(n)]
        oa = hip.image_array([hip.out_image(o.data_ptr()) for o in outs])
        torch.cuda.synchronize()
        assert lib.uhdr_hip_apply_gainmap_batch(n, ya, ma, C.byref(md), hip.OUTPUT_HDR_HLG, FLT_MAX, oa, mode,
                                                C.c_void_p(stream.cuda_stream)) == 0
        torch.cuda.synchronize()
        res[mode] = [to_host(o) for o in outs]
    for a, b in zip(res[hip.APPLY_EXACT], res[hip.APPLY_EXACT_UNFILTERED]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("mode", ["fast", "exact"])
def test_an_image_taller_than_the_grid(hip, orc, mode):
    """65 540 rows: the per-pixel kernels' row dimension of the grid stops at 65 535 and the rest is reached by striding"""
    from tests.test_gpu_parity import _oracle_apply, _check_apply
    lib = hip.load()
    w, h, scale = 8, 65540, 2
    rng = np.random.RandomState(3)
    yuv = rng.randint(0, 256, w * h * 3 // 2).astype(np.uint8)
    gmap = rng.randint(0, 256, (h // scale, w // scale)).astype(np.uint8)
    maxb = np.float32(4.0)
    md = hip.metadata(maxb)
    ref = _oracle_apply(orc, yuv, w, h, gmap, maxb, hip.OUTPUT_HDR_HLG, FLT_MAX)
    if mode == "exact":
        assert np.array_equal(_apply(lib, hip, yuv, w, h, gmap, md, hip.OUTPUT_HDR_HLG, FLT_MAX, hip.APPLY_EXACT), ref)
    else:
        _check_apply(hip, hip.OUTPUT_HDR_HLG, _apply(lib, hip, yuv, w, h, gmap, md, hip.OUTPUT_HDR_HLG, FLT_MAX, hip.APPLY_FAST), ref, w, h)
