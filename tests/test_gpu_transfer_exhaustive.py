"""-m gpu: exhaustive proofs for generate's lean double-precision transfer functions.

k_generate does not call ocml's f64 pow/exp per pixel; it evaluates a short f64 polynomial, applies a
Ziv rounding test and only falls back to the exact path when the float rounding is in doubt
(csrc/uhdr_device_math.h).  Bit-exactness therefore rests on two facts, both checked here:

 (1) for EVERY float input of the domain, the guarded function returns the same float as the exact
     (ocml double) evaluation of the reference's formula -- exhaustive, ~1e9 inputs per function;
 (2) the exact device evaluation equals the oracle (glibc libm) -- dense random samples + edge values.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _eval(lib, hip, fn, x, mn=1.0, mx=4.0):
    out = torch.empty_like(x)
    rc = lib.uhdr_hip_eval_transfer(fn, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), x.numel(), mn, mx,
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    return out


def _all_floats(lo_bits, hi_bits, chunk=1 << 27):
    for b in range(lo_bits, hi_bits + 1, chunk):
        n = min(chunk, hi_bits + 1 - b)
        yield (torch.arange(n, dtype=torch.int32, device="cuda") + b).view(torch.float32)


@pytest.mark.parametrize("fn,name", [(0, "srgbInvOetf"), (1, "hlgInvOetf"), (2, "pqInvOetf")])
def test_guarded_equals_exact_for_every_float_in_0_1(hip, fn, name):
    lib = hip.load()
    total = bad = 0
    for x in _all_floats(0, 0x3F800000):          # every float in [0, 1]
        a = _eval(lib, hip, fn, x).view(torch.int32)
        b = _eval(lib, hip, fn + 10, x).view(torch.int32)
        bad += int((a != b).sum().item())
        total += x.numel()
    assert total == 0x3F800001 and bad == 0, "%s: %d of %d inputs differ from the exact path" % (name, bad, total)


@pytest.mark.parametrize("fn,name", [(4, "hlgOetf"), (5, "pqOetf")])
def test_guarded_forward_oetf_equals_exact_for_every_float_in_0_64(hip, fn, name):
    """apply's EXACT mode (k_apply_px): the lean forward OETFs return the float of the exact (ocml double) path for every float
    in [0, 64] -- [0, 1] is what a call produces, values above 1 appear when max_display_boost < maxContentBoost"""
    lib = hip.load()
    total = bad = 0
    for x in _all_floats(0, 0x42800000):
        a = _eval(lib, hip, fn, x).view(torch.int32)
        b = _eval(lib, hip, fn + 10, x).view(torch.int32)
        bad += int((a != b).sum().item())
        total += x.numel()
    assert total == 0x42800001 and bad == 0, "%s: %d of %d inputs differ from the exact path" % (name, bad, total)


def test_guarded_gain_factor_equals_exact_for_every_float_in_minus32_32(hip):
    """(float)exp2((double)x), applyGain's factor (gainmapmath.cpp:553), for every float with |x| <= 32 (log2 of a content boost
    stays far inside), plus the values outside that take the exact path by construction"""
    lib = hip.load()
    bad = total = 0
    for lo, hi in ((0, 0x42000000), (0x80000000 - (1 << 32), 0xC2000000 - (1 << 32))):
        for x in _all_floats(lo, hi):
            a = _eval(lib, hip, 6, x).view(torch.int32)
            b = _eval(lib, hip, 16, x).view(torch.int32)
            bad += int((a != b).sum().item())
            total += x.numel()
    z = torch.tensor([100.0, -100.0, 127.9, -126.5, -140.0, 200.0, float("inf"), float("-inf")], dtype=torch.float32, device="cuda")
    assert torch.equal(_eval(lib, hip, 6, z).view(torch.int32), _eval(lib, hip, 16, z).view(torch.int32))
    assert bad == 0, (bad, total)


@pytest.mark.parametrize("boosts", [(1.0, 1000.0 / 203.0), (1.0, 10000.0 / 203.0), (0.25, 4.0)])
def test_guarded_encode_gain_equals_exact(hip, boosts):
    lib = hip.load()
    mn, mx = float(np.float32(boosts[0])), float(np.float32(boosts[1]))
    lo_bits = int(np.float32(mn / 4).view(np.uint32))
    hi_bits = int(np.float32(mx * 4).view(np.uint32))
    bad = total = 0
    for x in _all_floats(lo_bits, hi_bits):
        a, b = _eval(lib, hip, 3, x, mn, mx), _eval(lib, hip, 13, x, mn, mx)
        bad += int((a != b).sum().item())
        total += x.numel()
    assert bad == 0, (bad, total)
    z = torch.tensor([0.0, -1.0, 1e-30, 3e38], dtype=torch.float32, device="cuda")
    assert torch.equal(_eval(lib, hip, 3, z, mn, mx), _eval(lib, hip, 13, z, mn, mx))


def test_lean_path_is_taken_almost_always(hip):
    lib = hip.load()
    x = torch.rand(1 << 24, device="cuda")
    for fn in (100, 101):
        frac = float(_eval(lib, hip, fn, x).mean().item())
        assert frac > 0.9995, (fn, frac)
        print("fn %d: lean f64 path accepted for %.6f of random inputs" % (fn, frac))


@pytest.mark.parametrize("fn,ofn", [(10, 0), (11, 1), (12, 2), (14, 4), (15, 5)])
def test_exact_device_path_equals_glibc_oracle(hip, orc, fn, ofn):
    """ocml f64 pow/exp/log rounded to float == glibc's, on 6M random + structured inputs per function"""
    lib = hip.load()
    rng = np.random.RandomState(fn)
    xs = np.concatenate([rng.uniform(0, 1, 4_000_000), rng.uniform(0, 0.1, 1_000_000), np.arange(0, 1024) / 1023.0,
                         np.arange(0, 256) / 255.0, 10.0 ** rng.uniform(-6, 0, 1_000_000),
                         [0, 1, 0.04045, 0.5, 1 / 12, 1e-4]]).astype(np.float32)
    got = _eval(lib, hip, fn, torch.from_numpy(xs).cuda()).cpu().numpy()
    want = orc.eval_transfer(ofn, xs, threads=16)
    nbad = int((got.view(np.uint32) != want.view(np.uint32)).sum())
    assert nbad == 0, "%d of %d differ from glibc" % (nbad, xs.size)


def test_guarded_encode_gain_equals_glibc_oracle(hip, orc):
    lib = hip.load()
    rng = np.random.RandomState(9)
    for mn, mx in ((1.0, float(np.float32(1000.0) / np.float32(203.0))), (1.0, float(np.float32(10000.0) / np.float32(203.0)))):
        xs = np.concatenate([rng.uniform(0.5, mx * 1.2, 3_000_000), 2.0 ** rng.uniform(-2, 6, 1_000_000),
                             [mn, mx, 1.0, 2.0, 4.0]]).astype(np.float32)
        got = _eval(lib, hip, 3, torch.from_numpy(xs).cuda(), mn, mx).cpu().numpy()
        want = orc.eval_transfer(3, xs, mn, mx, threads=16)
        assert np.array_equal(got, want)


def test_map_byte_to_float_is_exact(hip):
    """gain-map byte / 255.0f (gainmapmath.cpp:632) through the 3-instruction constant division used by the FAST
    apply kernel equals the IEEE division for all 256 bytes"""
    lib = hip.load()
    x = torch.arange(256, dtype=torch.float32, device="cuda")
    assert torch.equal(_eval(lib, hip, 30, x).view(torch.int32), _eval(lib, hip, 31, x).view(torch.int32))
    want = (np.arange(256, dtype=np.float32) / np.float32(255.0)).view(np.uint32)
    assert np.array_equal(_eval(lib, hip, 31, x).cpu().numpy().view(np.uint32), want)
