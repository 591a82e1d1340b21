"""-m gpu: FAST apply's line-segment tables (csrc/uhdr_kernels.hip: tab_pair / code_eval2; csrc/uhdr_capi.hip: build_line_tables).

k_apply_s4 evaluates a channel as  code = stage2(T(c) * 2^(g E))  with T = srgbInvOetf^g read from a table whose cell is the
half-precision bit pattern of c, and stage2 = trunc(1023 * OETF) as a function of u = linear^g read from 129 uniform cells
(g = 1/2 for HLG, m1 for PQ; uhdr_kernels.h).  Measured here with the kernel's own index arithmetic (uhdr_hip_eval_transfer codes
50-54 read the device copy of the tables):

 * stage 1, every float in [0, 1]: relative error against srgbInvOetf^g (the exact device function, pinned to glibc in
   test_gpu_transfer_exhaustive.py, raised to g in double);
 * stage 2, every float in [0, 1]: the code against trunc(1023 * OETF(u^(1/g))) in double -- never more than one apart, and apart
   only where the exact code value lies within 0.06 of an integer.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

M1 = float(np.float32(2610.0) / np.float32(16384.0))
M2 = float(np.float32(2523.0) / np.float32(4096.0) * np.float32(128.0))
K1 = float(np.float32(3424.0) / np.float32(4096.0))
K2 = float(np.float32(2413.0) / np.float32(4096.0) * np.float32(32.0))
K3 = float(np.float32(2392.0) / np.float32(4096.0) * np.float32(32.0))
HA, HB, HC = float(np.float32(0.17883277)), float(np.float32(0.28466892)), float(np.float32(0.55991073))


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


@pytest.mark.parametrize("fn,g,name,tol", [(50, 1.0, "linear", 2e-5), (53, 0.5, "HLG (sqrt)", 1e-4), (54, M1, "PQ (^m1)", 1e-4)])
def test_stage1_table_for_every_float_in_0_1(hip, fn, g, name, tol):
    lib = hip.load()
    worst_rel = worst_abs = worst_sub = 0.0
    # g = 1 (cells = half-precision patterns): below 2^-17 the red and blue channels only ever see 0 ((y + k dv) / 255 has no
    # smaller positive value) and what green can see there is worth less than 0.01 code; below 2^-14 the cells are the half-precision
    # subnormals (uniform, 2^-21 wide).  g < 1 (cells = the value's own exponent and mantissa bits): every input the kernel can see,
    # i.e. from 2^-31 up (a channel is a sum of operands of magnitude >= 2^-8: a non-zero result is a multiple of 2^-31); the cell
    # of 0 doubles as the first sixteenth of that octave and is a line through the origin there.  0 itself is checked apart.
    sub = int(np.float32(2.0 ** -14).view(np.uint32))
    first = 2.0 ** -17 if g == 1.0 else 2.0 ** -31 * (1.0 + 1.0 / 16.0)
    for x in _all_floats(int(np.float32(first).view(np.uint32)), 0x3F800000):
        t = _eval(lib, fn, x).double()
        e = _eval(lib, 10, x).double().pow(g)
        d = (t - e).abs()
        worst_abs = max(worst_abs, float(d.max().item()))
        rel = d / e
        lo = x.view(torch.int32) < sub
        if bool(lo.any()):
            worst_sub = max(worst_sub, float(rel[lo].max().item()))
        if bool((~lo).any()):
            worst_rel = max(worst_rel, float(rel[~lo].max().item()))
    z = torch.zeros(4, dtype=torch.float32, device="cuda")
    assert float(_eval(lib, fn, z).abs().max().item()) == 0.0
    print("stage 1, %s: max |error| %.3g absolute, %.3g relative (%.3g below 2^-14)" % (name, worst_abs, worst_rel, worst_sub))
    # g = 1: 128 cells per octave; g < 1: 16 cells per octave, a constant relative error (6e-5 = 0.03 codes at most)
    assert worst_rel <= tol and worst_abs <= 1e-4 and worst_sub <= 1e-4, (worst_abs, worst_rel, worst_sub)


def _hlg_code(u):
    x = u * u
    return 1023.0 * torch.where(x <= 1.0 / 12.0, (3.0 * x).sqrt(), HA * (12.0 * x - HB).clamp_min(1e-300).log() + HC)


def _pq_code(p):
    return torch.where(p <= 0, torch.zeros_like(p), 1023.0 * ((K1 + K2 * p) / (1.0 + K3 * p)).pow(M2))


@pytest.mark.parametrize("fn,code,name", [(51, _hlg_code, "hlgOetf"), (52, _pq_code, "pqOetf")])
def test_stage2_code_table_for_every_float_in_0_1(hip, fn, code, name):
    lib = hip.load()
    ndiff = total = 0
    worst = 0
    far = 0.0
    for u in _all_floats(0, 0x3F800000):
        got = _eval(lib, fn, u).double()
        ev = code(u.double())
        want = ev.floor()
        d = (got - want).abs()
        worst = max(worst, int(d.max().item()))
        m = d != 0
        ndiff += int(m.sum().item())
        total += u.numel()
        if bool(m.any()):
            frac = ev[m] - ev[m].floor()
            far = max(far, float(torch.minimum(frac, 1.0 - frac).max().item()))
    print("%s stage 2: %d of %d floats give another code (%.2e), worst %d, farthest from an integer %.4f"
          % (name, ndiff, total, ndiff / total, worst, far))
    assert worst <= 1 and far <= 0.06, (worst, far)


@pytest.mark.parametrize("fmt", [2, 3])
def test_capped_display_boost_on_either_side_of_1(hip, orc, fmt):
    """max_display_boost < maxContentBoost: the table path serves the call as long as the largest factor it can produce,
    maxBoost^(display / max) / display, stays at or below 1 (launch_apply_t); above, channels pass 1.0, wrap through the reference's
    & 0x3ff and the special-function form runs.  For maxContentBoost 4.926 the two meet at a display boost of ~1.78: white pixels
    under gain 1 sit exactly on that maximum, on both sides of it"""
    from tests.gpu_util import gpu_apply, to_dev, diff_1010102
    from tests.test_gpu_parity import _oracle_apply
    lib = hip.load()
    w, h = 256, 64
    _, yuv = orc.lcg_frame(w, h, 21)
    yuv = yuv.copy()
    yuv[: w * 16] = 255                                   # 16 white rows ...
    yuv[w * h: w * h + (w // 2) * 4] = 128                # ... with neutral chroma
    yuv[w * h + (w // 2) * (h // 2): w * h + (w // 2) * (h // 2) + (w // 2) * 4] = 128
    gmap = np.random.RandomState(5).randint(0, 256, (h // 4, w // 4)).astype(np.uint8)
    gmap[:6] = 255
    maxb = np.float32(1000.0 / 203.0)
    dy, dmap = to_dev(yuv), to_dev(gmap)
    yi = hip.yuv420_image(dy.data_ptr(), w, h, hip.CG_BT709)
    md = hip.metadata(maxb)
    for boost in (1.0, 1.3, 1.7, 1.77, 1.78, 1.785, 1.79, 1.8, 1.9, 2.5, 4.0, 4.9):
        ref = _oracle_apply(orc, yuv, w, h, gmap, maxb, fmt, boost)
        st, fast, _ = gpu_apply(lib, yi, dmap, w // 4, h // 4, md, fmt, boost, hip.APPLY_FAST)
        assert st == 0
        worst, frac, ok = diff_1010102(fast.view(np.uint32), ref.view(np.uint32), wrap=True)
        assert ok and worst <= 1, (boost, worst)
        # and without the wrap allowance wherever the reference itself stays below 1024: no channel may differ by more than 1
        top = float(maxb) ** (boost / float(maxb)) / boost
        if top <= 1.0:
            worst, frac, ok = diff_1010102(fast.view(np.uint32), ref.view(np.uint32), wrap=False)
            assert worst <= 1, (boost, worst)
