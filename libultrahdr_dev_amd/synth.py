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


def lcg_frame(w, h, seed, device="cuda"):
    """-> (p010_bytes uint8[w*h*3], yuv uint8[w*h*3/2]) on `device`; p010 is little-endian uint16"""
    n_l, n_c = w * h, w * h // 2
    n = n_l + n_c
    i = torch.arange(n, dtype=torch.int64, device=device)
    d_p = _lcg_states(seed, 2 * i + 1) >> 8     # draws alternate: p010[i] then yuv[i]
    d_y = _lcg_states(seed, 2 * i + 2) >> 8
    mod = torch.where(i < n_l, 877, 897)
    p = ((64 + d_p % mod) << 6).to(torch.int32)
    p010 = torch.stack([(p & 0xFF).to(torch.uint8), (p >> 8).to(torch.uint8)], dim=1).reshape(-1)
    yuv = (d_y & 255).to(torch.uint8)
    return p010.contiguous(), yuv.contiguous()
