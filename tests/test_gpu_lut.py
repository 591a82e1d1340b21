"""-m gpu: the opt-in LUT mode.

The reference exports the five static transfer-function tables with their accessors (gainmapmath.cpp:21-64,
162-171, 269-354) and GainLUT (gainmapmath.h:149-182), and tests them (gainmapmath_test.cpp:808-939); its
generate / apply loops take them when the USE_*_LUT macros are visible (ultrahdr.cpp:230,238,319,433,446,470,481).
The oracle restatement of all of this is pinned to the reference's object code by tests/test_oracle_pins.py.
Here the device side is held to it: tables, accessors and whole pipelines, BIT-EXACT (a LUT pipeline is float
and integer arithmetic plus table reads; no tolerance applies).
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FLT_MAX = 3.4028234663852886e38
TABLES = ((0, 1024, 40, 10), (1, 4096, 41, 11), (2, 4096, 42, 12), (4, 65536, 44, 14), (5, 65536, 45, 15))  # which, N, lut fn, exact fn


def _eval(lib, fn, x, mn=1.0, mx=4.0):
    out = torch.empty_like(x)
    rc = lib.uhdr_hip_eval_transfer(fn, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), x.numel(), mn, mx,
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    return out


def _table(lib, which, n):
    buf = (C.c_float * n)()
    cnt = C.c_size_t()
    assert lib.uhdr_hip_lut_table(which, buf, n, C.byref(cnt)) == 0 and cnt.value == n
    return np.frombuffer(buf, np.float32).copy()


@pytest.mark.parametrize("which,n,fn,exact_fn", TABLES)
def test_static_tables_equal_the_oracle_and_the_direct_functions(hip, orc, which, n, fn, exact_fn):
    lib = hip.load()
    t = _table(lib, which, n)
    assert np.array_equal(t.view(np.uint32), orc.lut_table(which).view(np.uint32))
    # gainmapmath_test.cpp:808-841: f(idx/(N-1)) == fLUT(idx/(N-1)) at every knot
    knots = torch.from_numpy((np.arange(n, dtype=np.float32) / np.float32(n - 1)).astype(np.float32)).cuda()
    assert torch.equal(_eval(lib, fn, knots).view(torch.int32), _eval(lib, exact_fn, knots).view(torch.int32))
    small = (C.c_float * 4)()
    cnt = C.c_size_t()
    assert lib.uhdr_hip_lut_table(which, small, 4, C.byref(cnt)) == hip.ERROR_INSUFFICIENT_RESOURCE and cnt.value == n
    assert lib.uhdr_hip_lut_table(3, small, 4, C.byref(cnt)) == hip.ERROR_UNSUPPORTED_FEATURE


@pytest.mark.parametrize("which,n,fn,exact_fn", TABLES)
def test_accessor_index_rounding_for_every_float_in_0_2(hip, which, n, fn, exact_fn):
    """idx = uint32(double(e * (N-1)) + 0.5) clipped to N-1, checked for EVERY float in [0, 2] against the same
    expression evaluated by torch in float64"""
    lib = hip.load()
    table = torch.from_numpy(_table(lib, which, n)).cuda()
    bad = total = 0
    chunk = 1 << 27
    for b in range(0, 0x40000000 + 1, chunk):
        cnt = min(chunk, 0x40000000 + 1 - b)
        x = (torch.arange(cnt, dtype=torch.int32, device="cuda") + b).view(torch.float32)
        got = _eval(lib, fn, x)
        pos = (x * float(n - 1)).double() + 0.5            # float product (torch keeps float32), double sum
        idx = pos.floor().clamp_(max=n - 1).long()
        bad += int((got.view(torch.int32) != table[idx].view(torch.int32)).sum().item())
        total += cnt
    assert total == 0x40000001 and bad == 0, (bad, total)


def test_one_instruction_index_equals_the_reference_expression_for_every_float_below_2_31(hip):
    """The accessors form `static_cast<uint32_t>(e * (N - 1) + 0.5)` (float product, double sum, truncation: gainmapmath.cpp:162-171)
    with v_cvt_rpi_i32_f32 -- nearest integer, ties toward +infinity, no intermediate rounding.  Every float in [+0, 2^31) and -0
    through both forms inside the library (eval 47), and a second time against torch's float64 arithmetic on the indices read back
    through the gain table's accessor for the floats around every half-integer up to 2^24, where a sum rounded in float would slip."""
    lib = hip.load()
    total = 0
    chunk = 1 << 27
    for b in range(0, 0x4F000000, chunk):
        cnt = min(chunk, 0x4F000000 - b)
        x = (torch.arange(cnt, dtype=torch.int32, device="cuda") + b).view(torch.float32)
        ok = _eval(lib, 47, x)
        assert float(ok.min().item()) == 1.0, hex(b + int(ok.argmin().item()))
        total += cnt
    assert total == 0x4F000000
    assert float(_eval(lib, 47, torch.tensor([-0.0], device="cuda")).item()) == 1.0
    # the neighbours of k + 0.5 (the tie itself rounds up; its lower neighbour must not)
    k = torch.arange(0, 1 << 22, dtype=torch.float64, device="cuda") + 0.5
    for step in (-1, 0, 1):
        x = (k.float().view(torch.int32) + step).view(torch.float32)
        assert float(_eval(lib, 47, x).min().item()) == 1.0, step


def test_accessors_and_gain_factor_equal_the_oracle_including_out_of_range_inputs(hip, orc):
    lib = hip.load()
    rng = np.random.RandomState(5)
    xs = np.concatenate([rng.uniform(0, 1, 2_000_000), rng.uniform(0.99, 1.5, 100_000), 10.0 ** rng.uniform(-7, 0, 200_000),
                         [0, 1, 0.5, 2.0, 65535.0, 70000.0, 4.2949673e9, 4.3e9, 1e15, 1e19, 3e38,
                          -0.0, -1e-4, -4e-4, -6e-4, -0.3, -5.0, -1e19, -3e38, np.inf, -np.inf, np.nan]]).astype(np.float32)
    dx = torch.from_numpy(xs).cuda()
    for fn in (40, 41, 42, 44, 45):
        got = _eval(lib, fn, dx).cpu().numpy()
        want = orc.eval_transfer(fn, xs, threads=16)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), fn
    for mn, mx in ((1.0, 4.926108), (0.5, 8.0), (1.0, 49.26108), (0.25, 4.0)):
        got = _eval(lib, 46, dx, mn, mx).cpu().numpy()
        want = orc.eval_transfer(46, xs, mn, mx, threads=16)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (mn, mx)


def test_gain_lut_tables(hip, orc):                                               # gainmapmath_test.cpp:843-939
    lib = hip.load()
    out = (C.c_float * 1024)()
    cases = []
    for boost in range(1, 11):
        cases += [(1.0 / boost, float(boost)), (1.0, float(boost)), (float(1.0 / np.float32(boost) ** np.float32(1.0 / 3.0)), float(boost))]
    cases += [(1.0, 1000.0 / 203.0), (1.0, 10000.0 / 203.0), (0.5, 6.0)]
    for mn, mx in cases:
        mn, mx = float(np.float32(mn)), float(np.float32(mx))
        md = hip.metadata(mx, mn)
        assert lib.uhdr_hip_gain_lut(C.byref(md), 0, 0.0, out) == 0
        plain = np.frombuffer(out, np.float32).copy()
        assert np.array_equal(plain.view(np.uint32), orc.gain_lut("orc_", mn, mx).view(np.uint32)), (mn, mx)
        assert lib.uhdr_hip_gain_lut(C.byref(md), 1, mx, out) == 0
        assert np.array_equal(np.frombuffer(out, np.float32), plain)             # EXPECT_RGB_EQ: boost == max changes nothing
        for db in (1.0, 2.5, FLT_MAX, 0.0):
            assert lib.uhdr_hip_gain_lut(C.byref(md), 1, db, out) == 0
            assert np.array_equal(np.frombuffer(out, np.float32).view(np.uint32), orc.gain_lut("orc_", mn, mx, db).view(np.uint32)), (mn, mx, db)


def _pair(hip, orc, w, h, seed, sg, hg):
    from tests.gpu_util import to_dev
    p010, yuv = orc.lcg_frame(w, h, seed)
    dp, dy = to_dev(p010), to_dev(yuv)
    return (p010, yuv, dp, dy), hip.yuv420_image(dy.data_ptr(), w, h, sg), hip.p010_image(dp.data_ptr(), w, h, hg)


def _gpu_generate_lut(lib, hip, yi, pi, tf, is601=False):
    from tests.gpu_util import dev_empty, stream_ptr, to_host
    mw, mh = yi.width // 4, yi.height // 4
    dmap = dev_empty(mw * mh, 0xCD)
    dest = hip.out_image(dmap.data_ptr())
    md = hip.Metadata()
    st = lib.uhdr_hip_generate_gainmap_ex(C.byref(yi), C.byref(pi), tf, C.byref(md), C.byref(dest), int(is601),
                                          hip.GENERATE_LUT, hip.MEM_DEVICE, stream_ptr())
    return st, to_host(dmap, mw * mh).reshape(mh, mw), md, dmap


@pytest.mark.parametrize("tf", [0, 1, 2])
@pytest.mark.parametrize("dims", [(256, 128), (200, 100), (36, 20), (8, 4)])
def test_generate_lut_is_bit_exact(hip, orc, tf, dims):
    lib = hip.load()
    w, h = dims
    for sg, hg, is601 in ((0, 2, False), (1, 1, False), (2, 0, True)):
        keep, yi, pi = _pair(hip, orc, w, h, 100 + tf, sg, hg)
        st, m, md, _ = _gpu_generate_lut(lib, hip, yi, pi, tf, is601)
        oy, op = orc.yuv420_image(keep[1], w, h, sg), orc.p010_image(keep[0], w, h, hg)
        ost, om, omd = orc.generate("orc_", oy, op, tf, is601, threads=4, lut=True)
        assert st == ost == 0 and np.array_equal(m, om), (dims, tf, sg, hg, int((m != om).sum()))
        assert md.maxContentBoost == omd.maxContentBoost and md.version == b"1.0"


@pytest.mark.parametrize("fmt", [1, 2, 3, 4])
@pytest.mark.parametrize("boost", [FLT_MAX, 2.0])
def test_apply_lut_is_bit_exact(hip, orc, fmt, boost):
    from tests.gpu_util import gpu_apply, to_dev
    lib = hip.load()
    for (w, h, scale), (maxb, minb) in zip(((256, 128, 4), (96, 64, 2), (60, 36, 1), (64, 32, 8)),
                                          ((1000.0 / 203.0, 1.0), (10000.0 / 203.0, 1.0), (6.0, 0.5), (4.0, 0.25))):
        _, yuv = orc.lcg_frame(w, h, 200 + fmt)
        mw, mh = w // scale, h // scale
        gmap = np.random.RandomState(fmt).randint(0, 256, (mh, mw)).astype(np.uint8)
        dy, dmap = to_dev(yuv), to_dev(gmap)
        yi = hip.yuv420_image(dy.data_ptr(), w, h, hip.CG_BT709)
        md = hip.metadata(np.float32(maxb), np.float32(minb))
        omd = orc.Metadata(float(np.float32(maxb)), float(np.float32(minb)), 1.0, 0.0, 0.0, float(np.float32(minb)), float(np.float32(maxb)), 1)
        st, got, dest = gpu_apply(lib, yi, dmap, mw, mh, md, fmt, boost, hip.APPLY_LUT)
        ost, want, _ = orc.apply("orc_", orc.yuv420_image(yuv, w, h, orc.CG_BT709), gmap, omd, fmt, boost, threads=4, lut=True)
        assert st == ost == 0 and (dest.width, dest.height) == (w, h)
        assert np.array_equal(got, want), (fmt, boost, scale, int((got != want).sum()))


@pytest.mark.parametrize("fmt", [1, 3, 4])
def test_apply_lut_division_by_the_display_boost_over_many_divisors(hip, orc, fmt):
    """Round 4: interior waves of the scale-4 LUT kernel divide by display_boost with the IEEE expansion run WITHOUT its operand
    scaling, two quotients per instruction (lut_cell_pk), where the host has bounded the operands (AppConsts::lut_plain_div); other
    calls keep hipcc's full expansion.  Both must be the oracle's true division, bit for bit: a sweep of divisors -- round and
    awkward ones, significands of all ones and of one, below 1, large, and calls on which the guard switches the plain form off
    (divisor beyond 2^20, a quotient beyond 32768, minContentBoost above maxContentBoost) -- on a frame wide enough for interior waves."""
    from tests.gpu_util import gpu_apply, to_dev
    lib = hip.load()
    w, h = 1024, 256
    _, yuv = orc.lcg_frame(w, h, 900 + fmt)
    gmap = np.random.RandomState(50 + fmt).randint(0, 256, (h // 4, w // 4)).astype(np.uint8)
    dy, dmap = to_dev(yuv), to_dev(gmap)
    yi = hip.yuv420_image(dy.data_ptr(), w, h, hip.CG_BT709)
    ones = float(np.frombuffer(np.uint32(0x40FFFFFF).tobytes(), np.float32)[0])       # 7.9999995: a significand of all ones
    cases = [(4.9261084, 1.0, b) for b in (FLT_MAX, 1.0, 1.5, 2.0, 3.1415927, 4.0, 4.5, ones / 2, 1.0000001, 0.75, 0.001)]
    cases += [(49.261086, 1.0, b) for b in (FLT_MAX, 7.77, ones, 33.333332, 48.0)]
    cases += [(64.0, 0.25, FLT_MAX), (8.0, 0.5, 3.0), (1.0e6, 1.0, 2.0e6), (3.0e6, 1.0, FLT_MAX), (1000.0, 1.0, 0.01), (2.0, 8.0, FLT_MAX), (16.0, 1.0, 1.0e-5)]
    for maxb, minb, boost in cases:
        md = hip.metadata(np.float32(maxb), np.float32(minb))
        omd = orc.Metadata(float(np.float32(maxb)), float(np.float32(minb)), 1.0, 0.0, 0.0, float(np.float32(minb)), float(np.float32(maxb)), 1)
        st, got, _ = gpu_apply(lib, yi, dmap, w // 4, h // 4, md, fmt, boost, hip.APPLY_LUT)
        ost, want, _ = orc.apply("orc_", orc.yuv420_image(yuv, w, h, orc.CG_BT709), gmap, omd, fmt, boost, threads=8, lut=True)
        assert st == ost == 0
        assert np.array_equal(got, want), (fmt, maxb, minb, boost, int((got != want).sum()))


def test_lut_modes_through_host_entry_points_and_batches(hip, orc):
    """MEM_HOST single calls and a mixed-size batch take the same LUT kernels"""
    from tests.gpu_util import dev_empty, stream_ptr, to_dev, to_host
    lib = hip.load()
    w, h = 128, 64
    p010, yuv = orc.lcg_frame(w, h, 321)
    yi = hip.yuv420_image(yuv.ctypes.data, w, h, hip.CG_BT709)
    pi = hip.p010_image(p010.ctypes.data, w, h, hip.CG_BT2100)
    m = np.zeros((h // 4, w // 4), np.uint8)
    dest = hip.out_image(m.ctypes.data)
    md = hip.Metadata()
    assert lib.uhdr_hip_generate_gainmap_ex(C.byref(yi), C.byref(pi), hip.TF_HLG, C.byref(md), C.byref(dest), 0, hip.GENERATE_LUT,
                                            hip.MEM_HOST, None) == 0
    ost, om, omd = orc.generate("orc_", orc.yuv420_image(yuv, w, h, 0), orc.p010_image(p010, w, h, 2), 1, threads=4, lut=True)
    assert np.array_equal(m, om) and (dest.width, dest.height) == (w // 4, h // 4)
    out = np.zeros(w * h * 4, np.uint8)
    d2 = hip.out_image(out.ctypes.data)
    mi = hip.mono_image(m.ctypes.data, w // 4, h // 4)
    assert lib.uhdr_hip_apply_gainmap(C.byref(yi), C.byref(mi), C.byref(md), hip.OUTPUT_HDR_HLG, FLT_MAX, C.byref(d2), hip.APPLY_LUT,
                                      hip.MEM_HOST, None) == 0
    _, want, _ = orc.apply("orc_", orc.yuv420_image(yuv, w, h, 0), om, omd, orc.OUT_HDR_HLG, FLT_MAX, threads=4, lut=True)
    assert np.array_equal(out, want)
    assert lib.uhdr_hip_apply_gainmap(C.byref(yi), C.byref(mi), C.byref(md), hip.OUTPUT_HDR_HLG, FLT_MAX, C.byref(d2), 7,
                                      hip.MEM_HOST, None) == hip.ERROR_UNSUPPORTED_FEATURE
    assert lib.uhdr_hip_generate_gainmap_ex(C.byref(yi), C.byref(pi), hip.TF_HLG, C.byref(md), C.byref(dest), 0, 5,
                                            hip.MEM_HOST, None) == hip.ERROR_UNSUPPORTED_FEATURE

    # batch of three images, two sizes
    sizes = [(64, 32), (64, 32), (48, 24)]
    keep, yis, pis, dests, maps = [], [], [], [], []
    for i, (bw, bh) in enumerate(sizes):
        bp, by = orc.lcg_frame(bw, bh, 400 + i)
        dp, dy, dm = to_dev(bp), to_dev(by), dev_empty(bw * bh // 16, 0xCD)
        keep.append((bp, by, dp, dy, dm))
        yis.append(hip.yuv420_image(dy.data_ptr(), bw, bh, hip.CG_BT709))
        pis.append(hip.p010_image(dp.data_ptr(), bw, bh, hip.CG_BT2100))
        dests.append(hip.out_image(dm.data_ptr()))
    ya, pa, da = hip.image_array(yis), hip.image_array(pis), hip.image_array(dests)
    md = hip.Metadata()
    assert lib.uhdr_hip_generate_gainmap_batch_ex(3, ya, pa, hip.TF_PQ, C.byref(md), da, 0, hip.GENERATE_LUT, None, stream_ptr()) == 0
    outs = [dev_empty(bw * bh * 4, 0xCD) for bw, bh in sizes]
    mimgs = hip.image_array([hip.mono_image(keep[i][4].data_ptr(), sizes[i][0] // 4, sizes[i][1] // 4) for i in range(3)])
    oa = hip.image_array([hip.out_image(o.data_ptr()) for o in outs])
    assert lib.uhdr_hip_apply_gainmap_batch(3, ya, mimgs, C.byref(md), hip.OUTPUT_HDR_PQ, FLT_MAX, oa, hip.APPLY_LUT, stream_ptr()) == 0
    for i, (bw, bh) in enumerate(sizes):
        oy, op = orc.yuv420_image(keep[i][1], bw, bh, 0), orc.p010_image(keep[i][0], bw, bh, 2)
        _, om, omd = orc.generate("orc_", oy, op, 2, threads=2, lut=True)
        assert np.array_equal(to_host(keep[i][4], bw * bh // 16).reshape(bh // 4, bw // 4), om), i
        _, want, _ = orc.apply("orc_", oy, om, omd, orc.OUT_HDR_PQ, FLT_MAX, threads=2, lut=True)
        assert np.array_equal(to_host(outs[i], bw * bh * 4), want), i


def test_lut_full_size_4k_checksums(hip, orc):
    """C2-sized frame: LUT generate + apply on the GPU against the oracle's LUT pipeline (16 threads)"""
    from tests.gpu_util import gpu_apply
    lib = hip.load()
    w, h = 3840, 2160
    keep, yi, pi = _pair(hip, orc, w, h, 1234, hip.CG_BT709, hip.CG_BT2100)
    st, m, md, dmap = _gpu_generate_lut(lib, hip, yi, pi, hip.TF_HLG)
    oy, op = orc.yuv420_image(keep[1], w, h, 0), orc.p010_image(keep[0], w, h, 2)
    ost, om, omd = orc.generate("orc_", oy, op, 1, threads=16, lut=True)
    assert st == ost == 0 and np.array_equal(m, om)
    st, got, _ = gpu_apply(lib, yi, dmap, w // 4, h // 4, md, hip.OUTPUT_HDR_HLG, FLT_MAX, hip.APPLY_LUT)
    ost, want, _ = orc.apply("orc_", oy, om, omd, orc.OUT_HDR_HLG, FLT_MAX, threads=16, lut=True)
    assert st == ost == 0 and np.array_equal(got, want)
