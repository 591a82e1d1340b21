#!/usr/bin/env python3
"""Measured HBM ceilings of this box (torch kernels; plumbing only): write-only fill, copy, read-only reduction.
Printed as GB/s of algorithmic bytes; bench.py reports the copy figure beside the 8 TB/s spec."""
import json
import torch


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def measure(nbytes=2 << 30):
    n = nbytes // 4
    a = torch.empty(n, dtype=torch.float32, device="cuda")
    b = torch.empty(n, dtype=torch.float32, device="cuda")
    a.fill_(1.0)
    out = {}
    t = timed(lambda: b.fill_(2.0))
    out["fill_GBs"] = nbytes / t / 1e9
    t = timed(lambda: b.copy_(a))
    out["copy_GBs"] = 2 * nbytes / t / 1e9
    t = timed(lambda: torch.sum(a))
    out["read_sum_GBs"] = nbytes / t / 1e9
    # 3 reads : 1 write and 1 read : 3 writes mixes (apply writes 72 % of its bytes, generate reads 99 %)
    c = torch.empty(n, dtype=torch.float32, device="cuda")
    t = timed(lambda: torch.add(a, b, out=c))
    out["triad_2r1w_GBs"] = 3 * nbytes / t / 1e9
    return out


if __name__ == "__main__":
    print(json.dumps(measure()))
