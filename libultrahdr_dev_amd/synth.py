"""Synthetic P010 + YUV420 frame pairs generated directly in HBM (torch is plumbing here).

Reproduces the deterministic LCG frames of SURVEY.md section 8(d) without a serial loop:
``s <- s*1664525 + 1013904223 (mod 2^32)``, ``draw = s >> 8``; the k-th state is obtained by
jump-ahead (composition of affine maps by binary exponentiation), so every element is independent.
Layout: p010 = uint16[w*h*3/2] (luma, then interleaved UV), yuv = uint8[w*h*3/2] (Y, U, V).
"""
import torch

_A, _C, _MASK = 1664525, 1013904223, 0xFFFFFFFF


def _lcg_states(seed, k):
    """states s_k (int64 tensor holding uint32 values) for step indices k >= 1 (int64 tensor)"""
    ar = torch.ones_like(k)
    cr = torch.zeros_like(k)
    ap, cp = _A, _C
    nbits = int(k.max().item()).bit_length()
    for b in range(nbits):
        bit = ((k >> b) & 1).bool()
        ar_n = (ar * ap) & _MASK
        cr_n = (cr * ap + cp) & _MASK
        ar = torch.where(bit, ar_n, ar)
        cr = torch.where(bit, cr_n, cr)
        cp = (cp * ap + cp) & _MASK
        ap = (ap * ap) & _MASK
    return (ar * (seed & _MASK) + cr) & _MASK


def lcg_frame(w, h, seed, device="cuda", out=None):
    """-> (p010_bytes uint8[w*h*3], yuv uint8[w*h*3/2]) on `device`; p010 is little-endian uint16.  On a GPU the frame is written
    by the library's own kernel (uhdr_hip_synth_lcg_frame); the torch form below serves CPU tensors (and is what the CPU tests
    compare with the oracle's serial loop)."""
    import os
    if str(device).startswith("cuda") and not os.environ.get("UHDR_SYNTH_TORCH"):
        import ctypes as C

        from . import api
        dev = torch.device(device)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        lib = api.init(idx)
        with torch.cuda.device(idx):
            if out is not None:   # (uint8 tensors of w*h*3 and w*h*3/2 bytes the caller placed)
                p010, yuv = out
                assert p010.numel() == w * h * 3 and yuv.numel() == w * h * 3 // 2 and p010.is_contiguous() and yuv.is_contiguous()
            else:
                p010 = torch.empty(w * h * 3, dtype=torch.uint8, device=dev)
                yuv = torch.empty(w * h * 3 // 2, dtype=torch.uint8, device=dev)
            rc = lib.uhdr_hip_synth_lcg_frame(w, h, seed & _MASK, C.c_void_p(p010.data_ptr()), C.c_void_p(yuv.data_ptr()),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError("uhdr_hip_synth_lcg_frame failed: %d" % rc)
        return p010, yuv
    n_l, n_c = w * h, w * h // 2
    n = n_l + n_c
    i = torch.arange(n, dtype=torch.int64, device=device)
    d_p = _lcg_states(seed, 2 * i + 1) >> 8     # draws alternate: p010[i] then yuv[i]
    d_y = _lcg_states(seed, 2 * i + 2) >> 8
    mod = torch.where(i < n_l, 877, 897)
    p = ((64 + d_p % mod) << 6).to(torch.int32)
    p010 = torch.stack([(p & 0xFF).to(torch.uint8), (p >> 8).to(torch.uint8)], dim=1).reshape(-1)
    yuv = (d_y & 255).to(torch.uint8)
    if out is not None:
        out[0].copy_(p010)
        out[1].copy_(yuv)
        return out
    return p010.contiguous(), yuv.contiguous()


def smooth_frame(w, h, seed, device="cuda"):
    """SURVEY.md 8(d) 'smooth' variant: low-frequency cosine planes in the legal ranges plus a little noise, built on the
    device.  Returns (p010 uint16[w*h*3/2], yuv uint8[w*h*3/2]) like lcg_frame.  Natural-image-like statistics matter for
    the JPEG timings (entropy-coded size depends on content); the pixel kernels do not care."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)

    def plane(pw, ph, lo, hi, noise):
        yy = torch.arange(ph, device=device, dtype=torch.float32).view(-1, 1)
        xx = torch.arange(pw, device=device, dtype=torch.float32).view(1, -1)
        acc = torch.zeros((ph, pw), device=device)
        for k in range(3):
            fx, fy, ph0 = (0.7 + 0.9 * k) / pw, (1.1 + 0.6 * k) / ph, 0.5 + 1.3 * k + 0.01 * seed
            acc = acc + torch.cos(6.283185 * (fx * xx + fy * yy) + ph0)
        v = lo + (acc / 6.0 + 0.5) * (hi - lo) + (torch.rand((ph, pw), generator=g, device=device) - 0.5) * noise
        return v.clamp_(lo, hi)

    y8 = plane(w, h, 0, 255, 6.0).to(torch.uint8).reshape(-1)
    u8 = plane(w // 2, h // 2, 16, 240, 3.0).to(torch.uint8).reshape(-1)
    v8 = plane(w // 2, h // 2, 16, 240, 3.0).flip(0).to(torch.uint8).reshape(-1)
    yuv = torch.cat([y8, u8, v8])
    hy = (plane(w, h, 64, 940, 12.0).to(torch.int32) << 6).to(torch.uint16).reshape(-1)
    hu = (plane(w // 2, h // 2, 64, 960, 6.0).to(torch.int32) << 6)
    hv = (plane(w // 2, h // 2, 64, 960, 6.0).flip(1).to(torch.int32) << 6)
    huv = torch.stack([hu, hv], dim=-1).to(torch.uint16).reshape(-1)
    return torch.cat([hy, huv]), yuv
