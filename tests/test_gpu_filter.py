"""-m gpu: the f32 pre-filter of generate (k_generate<..., FILTER>).

generate stays bit-exact because the filter only decides which waves may skip the double-precision path: a pixel's
byte is taken from the f32 evaluation only when its pre-truncation code value is farther from an integer than the
worst-case distance between the f32 and the exact evaluation.  This file checks
  (1) the per-function error bounds that budget rests on, for EVERY float of the domain;
  (2) that filtered and unfiltered kernels produce identical bytes AND identical (exact) content min/max on large and
      on adversarial content (dark, flat, saturated, smooth, code-boundary ramps), for all gamut pairs;
  (3) both against the CPU oracle on a sample.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SRGB_FAST_REL, HLG_FAST_REL, PQ_FAST_REL, LOG2_ABS = 6.0e-7, 4.0e-7, 3.4e-6, 5.0e-7   # what generate_consts' kRel / kLogAbs assume


def _eval(lib, fn, x):
    out = torch.empty_like(x)
    assert lib.uhdr_hip_eval_transfer(fn, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), x.numel(), 1.0, 4.0,
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    return out


def _bits(v):
    return int(np.float32(v).view(np.uint32))


@pytest.mark.parametrize("fast,exact,lo,bound,name", [(20, 10, 1e-12, SRGB_FAST_REL, "sRGB EOTF"), (21, 11, 1e-12, HLG_FAST_REL, "HLG inverse OETF"),
                                                      (22, 12, float(np.nextafter(np.float32(1e-4), np.float32(1))), PQ_FAST_REL, "PQ inverse OETF")])
def test_fast_transfer_functions_relative_error_for_every_float(hip, fast, exact, lo, bound, name):
    lib = hip.load()
    worst = 0.0
    for b in range(_bits(lo), 0x3F800000 + 1, 1 << 27):
        n = min(1 << 27, 0x3F800000 + 1 - b)
        x = (torch.arange(n, dtype=torch.int32, device="cuda") + b).view(torch.float32)
        a, e = _eval(lib, fast, x).double(), _eval(lib, exact, x).double()
        worst = max(worst, float(((a - e).abs() / e).max()))
    print("%s: worst relative error of the f32 form %.3e" % (name, worst))
    assert worst <= bound, worst
    # zero <-> zero (the "SDR luminance is zero" decision must agree on both paths), and nothing negative
    z = torch.tensor([0.0], device="cuda")
    assert float(_eval(lib, fast, z)) == 0.0 and float(_eval(lib, exact, z)) == 0.0
    tiny = (torch.arange(1, 1 << 20, dtype=torch.int32, device="cuda") * 977).view(torch.float32)   # denormals .. 1e-30
    assert bool((_eval(lib, fast, tiny) >= 0).all())
    if fast == 22:   # the PQ cut-off (gainmapmath.cpp:328): zero up to and including 1e-4 on both paths
        cut = (torch.arange(-2000, 1, dtype=torch.int32, device="cuda") + _bits(1e-4)).view(torch.float32)
        assert bool((_eval(lib, fast, cut) == 0).all()) and bool((_eval(lib, exact, cut) == 0).all())


def test_hardware_log2_absolute_error_on_the_gain_range(hip):
    lib = hip.load()
    worst = 0.0
    for b in range(_bits(0.25), _bits(64.0) + 1, 1 << 27):
        n = min(1 << 27, _bits(64.0) + 1 - b)
        x = (torch.arange(n, dtype=torch.int32, device="cuda") + b).view(torch.float32)
        worst = max(worst, float((_eval(lib, 23, x).double() - torch.log2(x.double())).abs().max()))
    print("v_log_f32: worst absolute error on [0.25, 64] %.3e" % worst)
    assert worst <= LOG2_ABS


def _frames(kind, w, h, n, rng):
    """n (p010, yuv) pairs of one content class, legal ranges (P010 10-bit code << 6)"""
    out = []
    for i in range(n):
        if kind == "random":
            y = rng.randint(0, 256, w * h * 3 // 2).astype(np.uint8)
            p = (rng.randint(64, 941, w * h * 3 // 2).astype(np.uint16) << 6)
        elif kind == "dark":      # luminances near zero: the relative-error regime of the fast path ends here
            y = np.concatenate([rng.randint(0, 4, w * h), rng.randint(126, 131, w * h // 2)]).astype(np.uint8)
            p = np.concatenate([rng.randint(64, 70, w * h), rng.randint(510, 515, w * h // 2)]).astype(np.uint16) << 6
        elif kind == "flat":      # every gain equal: the statistics pass sees nothing but candidates
            y = np.concatenate([np.full(w * h, 90 + i), np.full(w * h // 2, 128)]).astype(np.uint8)
            p = np.concatenate([np.full(w * h, 500 + 7 * i), np.full(w * h // 2, 512)]).astype(np.uint16) << 6
        elif kind == "black":     # SDR luminance exactly zero: gain := 1
            y = np.concatenate([np.zeros(w * h), np.full(w * h // 2, 128)]).astype(np.uint8)
            p = np.concatenate([rng.randint(64, 941, w * h), np.full(w * h // 2, 512)]).astype(np.uint16) << 6
        elif kind == "saturated":
            y = np.concatenate([rng.choice([0, 255], w * h), rng.choice([0, 255], w * h // 2)]).astype(np.uint8)
            p = np.concatenate([rng.choice([64, 940], w * h), rng.choice([64, 960], w * h // 2)]).astype(np.uint16) << 6
        else:                     # "ramp": slowly varying grey, sweeps the gain through every code boundary
            yy = (np.arange(w * h) * 255.0 / (w * h)).astype(np.uint8)
            y = np.concatenate([yy, np.full(w * h // 2, 128)]).astype(np.uint8)
            pp = (64 + (np.arange(w * h)[::-1] * 876.0 / (w * h) * (0.3 + 0.1 * i))).astype(np.uint16)
            p = np.concatenate([pp, np.full(w * h // 2, 512)]).astype(np.uint16) << 6
        out.append((p, y))
    return out


def _run(lib, hip, frames, w, h, sg, hg, tf, mode, is601=0):
    from tests.gpu_util import dev_empty, stream_ptr, to_dev, to_host
    n = len(frames)
    keep = [(to_dev(p), to_dev(y), dev_empty(w * h // 16, 0xCD)) for p, y in frames]
    ya = hip.image_array([hip.yuv420_image(k[1].data_ptr(), w, h, sg) for k in keep])
    pa = hip.image_array([hip.p010_image(k[0].data_ptr(), w, h, hg) for k in keep])
    da = hip.image_array([hip.out_image(k[2].data_ptr()) for k in keep])
    mm = torch.full((2 * n,), 7.0, dtype=torch.float32, device="cuda")
    md = hip.Metadata()
    assert lib.uhdr_hip_generate_gainmap_batch_ex(n, ya, pa, tf, C.byref(md), da, is601, mode, C.c_void_p(mm.data_ptr()), stream_ptr()) == 0
    maps = [to_host(k[2], w * h // 16).copy() for k in keep]
    # and once more without statistics (the kernel skips the bookkeeping)
    for k in keep:
        k[2].fill_(0xEE)
    assert lib.uhdr_hip_generate_gainmap_batch_ex(n, ya, pa, tf, C.byref(md), da, is601, mode, None, stream_ptr()) == 0
    maps2 = [to_host(k[2], w * h // 16).copy() for k in keep]
    return maps, maps2, mm.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("kind", ["random", "dark", "flat", "black", "saturated", "ramp"])
@pytest.mark.parametrize("tf", [0, 1, 2])
def test_filtered_equals_unfiltered_and_oracle(hip, orc, kind, tf):
    lib = hip.load()
    w, h, n = 512, 256, 6
    rng = np.random.RandomState(hash(kind) % 1000 + tf)
    frames = _frames(kind, w, h, n, rng)
    for sg, hg, is601 in ((0, 2, 0), (1, 0, 0), (2, 1, 1), (0, 0, 0)):
        fm, fm2, fstat = _run(lib, hip, frames, w, h, sg, hg, tf, hip.GENERATE_EXACT, is601)
        um, um2, ustat = _run(lib, hip, frames, w, h, sg, hg, tf, hip.GENERATE_UNFILTERED, is601)
        for i in range(n):
            assert np.array_equal(fm[i], um[i]) and np.array_equal(fm2[i], um[i]) and np.array_equal(um2[i], um[i]), (kind, tf, sg, hg, i)
        assert np.array_equal(fstat, ustat), (kind, tf, sg, hg)
        # the CPU oracle on the first two frames (bytes and the exact min / max of the unclamped gain)
        for i in range(2):
            oy, op = orc.yuv420_image(frames[i][1], w, h, sg), orc.p010_image(frames[i][0], w, h, hg)
            st, om, _, omm = orc.generate("orc_", oy, op, tf, bool(is601), threads=8, stats=True)
            assert st == 0 and np.array_equal(om.reshape(-1), fm[i]), (kind, tf, sg, hg, i)
            assert np.array_equal(np.array(omm, np.float32).view(np.uint32), fstat[2 * i:2 * i + 2]), (kind, i, omm)


@pytest.mark.parametrize("tf", [1, 2])
def test_filtered_equals_unfiltered_on_64_4k_frames(hip, tf):
    """BASELINE configs[2] size: 64 x 4K LCG pairs, 33 M map pixels, bytes and statistics (HLG and PQ)"""
    from libultrahdr_dev_amd import synth
    from tests.gpu_util import stream_ptr
    lib = hip.load()
    w, h, n = 3840, 2160, 64
    fr = [synth.lcg_frame(w, h, 9000 + i) for i in range(n)]
    res = {}
    for mode in (hip.GENERATE_EXACT, hip.GENERATE_UNFILTERED):
        maps = [torch.full((w * h // 16,), 0xCD, dtype=torch.uint8, device="cuda") for _ in range(n)]
        ya = hip.image_array([hip.yuv420_image(f[1].data_ptr(), w, h, hip.CG_BT709) for f in fr])
        pa = hip.image_array([hip.p010_image(f[0].data_ptr(), w, h, hip.CG_BT2100) for f in fr])
        da = hip.image_array([hip.out_image(m.data_ptr()) for m in maps])
        mm = torch.zeros(2 * n, dtype=torch.float32, device="cuda")
        md = hip.Metadata()
        assert lib.uhdr_hip_generate_gainmap_batch_ex(n, ya, pa, tf, C.byref(md), da, 0, mode, C.c_void_p(mm.data_ptr()), stream_ptr()) == 0
        torch.cuda.synchronize()
        res[mode] = (torch.stack(maps), mm.clone())
    a, b = res[hip.GENERATE_EXACT], res[hip.GENERATE_UNFILTERED]
    assert torch.equal(a[0], b[0]), int((a[0] != b[0]).sum())
    assert torch.equal(a[1].view(torch.int32), b[1].view(torch.int32))


@pytest.mark.parametrize("w,h,n", [(3840, 2160, 1), (1920, 1080, 3), (1280, 720, 20), (3840, 2160, 12), (1024, 512, 2)])
@pytest.mark.parametrize("kind", ["random", "flat", "ramp"])
def test_statistics_routes_of_small_and_medium_launches(hip, kind, w, h, n):
    """A launch with statistics takes one of three routes by its size (uhdr_capi.hip): the exact kernel (tiny), the filtered
    kernel with one span per block + k_generate_resolve, or the four-span kernel + resolve; launches of few large images publish
    their estimates per list (GenConsts::stat_spread).  Every route, twice in a row on one stream (the resolve kernel leaves the
    header -- per-list words included -- cleared), on content that fills the lists (flat: every pixel a candidate or in doubt)
    and content that does not: bytes and statistics of the unfiltered kernel."""
    lib = hip.load()
    rng = np.random.RandomState(w + n)
    frames = _frames(kind, w, h, n, rng)
    um, _, ustat = _run(lib, hip, frames, w, h, 0, 2, 1, hip.GENERATE_UNFILTERED)
    for _ in range(2):
        fm, fm2, fstat = _run(lib, hip, frames, w, h, 0, 2, 1, hip.GENERATE_EXACT)
        for i in range(n):
            assert np.array_equal(fm[i], um[i]) and np.array_equal(fm2[i], um[i]), (kind, w, h, n, i)
        assert np.array_equal(fstat, ustat), (kind, w, h, n)
