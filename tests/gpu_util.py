"""Helpers for the -m gpu parity tests: move numpy planes to HBM with torch (plumbing only) and call
the C-ABI with device pointers."""
import ctypes as C

import numpy as np
import torch

from libultrahdr_dev_amd import api


def to_dev(a):
    """numpy array -> uint8 cuda tensor holding the same bytes"""
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()


def dev_empty(nbytes, fill=None):
    t = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device="cuda")
    if fill is not None:
        t.fill_(fill)
    return t


def to_host(t, nbytes=None, dtype=np.uint8):
    torch.cuda.synchronize()
    a = t.cpu().numpy()
    if nbytes is not None:
        a = a[:nbytes]
    return a.view(dtype)


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def gpu_generate(lib, yuv_img, p010_img, tf, sdr_is_601=False, stats=False, batch=None):
    """single-image device-memory generate; returns (status, map ndarray, metadata[, (min,max)])"""
    w, h = yuv_img.width, yuv_img.height
    mw, mh = w // 4, h // 4
    dmap = dev_empty(mw * mh, 0xCD)
    dest = api.out_image(dmap.data_ptr())
    md = api.Metadata()
    if stats:
        mm = torch.zeros(2, dtype=torch.float32, device="cuda")
        st = lib.uhdr_hip_generate_gainmap_batch(1, C.byref(yuv_img), C.byref(p010_img), tf, C.byref(md),
                                                 C.byref(dest), int(sdr_is_601), C.c_void_p(mm.data_ptr()),
                                                 stream_ptr())
        return st, to_host(dmap, mw * mh).reshape(mh, mw), md, tuple(mm.cpu().tolist()), dest
    st = lib.uhdr_hip_generate_gainmap(C.byref(yuv_img), C.byref(p010_img), tf, C.byref(md), C.byref(dest),
                                       int(sdr_is_601), api.MEM_DEVICE, stream_ptr())
    return st, to_host(dmap, mw * mh).reshape(mh, mw), md, dest


def gpu_apply(lib, yuv_img, dmap_t, mw, mh, md, fmt, max_display_boost, mode=api.APPLY_FAST):
    w, h = yuv_img.width, yuv_img.height
    nbytes = api.output_bytes(fmt, w, h)
    dout = dev_empty(nbytes, 0xCD)
    mimg = api.mono_image(dmap_t.data_ptr(), mw, mh)
    dest = api.out_image(dout.data_ptr())
    st = lib.uhdr_hip_apply_gainmap(C.byref(yuv_img), C.byref(mimg), C.byref(md), fmt, max_display_boost,
                                    C.byref(dest), mode, api.MEM_DEVICE, stream_ptr())
    return st, to_host(dout, nbytes), dest


def diff_1010102(a_u32, b_u32, wrap=False):
    """per-channel absolute difference of two RGBA1010102 arrays -> (max_abs_diff, fraction_differing, alpha_ok).
    wrap: the call's max_display_boost was below maxContentBoost.  Only then can a channel land on 1024 and wrap to 0 through the
    reference's `& 0x3ff` (gainmapmath.cpp:722-727), and only then is the distance taken modulo 1024; otherwise a channel that
    is 0 on one side and 1023 on the other is 1023 apart."""
    a = a_u32.astype(np.int64)
    b = b_u32.astype(np.int64)
    worst, ndiff = 0, 0
    for sh in (0, 10, 20):
        d = np.abs(((a >> sh) & 0x3ff) - ((b >> sh) & 0x3ff))
        if wrap:
            d = np.minimum(d, 1024 - d)
        worst = max(worst, int(d.max()) if d.size else 0)
        ndiff += int((d != 0).sum())
    alpha_ok = bool((((a >> 30) & 3) == ((b >> 30) & 3)).all())
    return worst, ndiff / max(3 * a.size, 1), alpha_ok


def half_ulp_diff(a_u16, b_u16):
    """distance in half-precision ULPs between two arrays of non-negative finite halfs (bit patterns)"""
    d = np.abs(a_u16.astype(np.int32) - b_u16.astype(np.int32))
    return int(d.max()) if d.size else 0, float((d != 0).mean()) if d.size else 0.0
