#!/usr/bin/env python3
"""host <-> device copy rates of this box: pageable and page-locked host memory, one direction and both at once"""
import json
import time

import torch

n = 64 << 20
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
dev2 = torch.empty(n, dtype=torch.uint8, device="cuda")
pag, pin = torch.empty(n, dtype=torch.uint8), torch.empty(n, dtype=torch.uint8).pin_memory()
pin2 = torch.empty(n, dtype=torch.uint8).pin_memory()
pag.fill_(1); pin.fill_(2); pin2.fill_(3)
out = {}


def wall(fn, it=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it


out["h2d_pageable_GBs"] = n / wall(lambda: dev.copy_(pag)) / 1e9
out["h2d_pinned_GBs"] = n / wall(lambda: dev.copy_(pin, non_blocking=True)) / 1e9
out["d2h_pageable_GBs"] = n / wall(lambda: pag.copy_(dev)) / 1e9
out["d2h_pinned_GBs"] = n / wall(lambda: pin.copy_(dev, non_blocking=True)) / 1e9
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def both():
    with torch.cuda.stream(s1):
        dev.copy_(pin, non_blocking=True)
    with torch.cuda.stream(s2):
        pin2.copy_(dev2, non_blocking=True)


out["duplex_pinned_GBs_each_way"] = n / wall(both) / 1e9
# one direction split over k streams (k copy engines), pinned and pageable
for k in (2, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(k)]
    c = n // k
    for name, host in (("pinned", pin), ("pageable", pag)):
        def h2d():
            for i, st in enumerate(streams):
                with torch.cuda.stream(st):
                    dev[i * c:(i + 1) * c].copy_(host[i * c:(i + 1) * c], non_blocking=True)

        def d2h():
            for i, st in enumerate(streams):
                with torch.cuda.stream(st):
                    host[i * c:(i + 1) * c].copy_(dev[i * c:(i + 1) * c], non_blocking=True)

        out["h2d_%s_%d_streams_GBs" % (name, k)] = n / wall(h2d) / 1e9
        out["d2h_%s_%d_streams_GBs" % (name, k)] = n / wall(d2h) / 1e9
t0 = time.perf_counter()
for _ in range(5):
    pin.copy_(pag)
out["host_memcpy_pageable_to_pinned_GBs_1_thread"] = 5 * n / (time.perf_counter() - t0) / 1e9
print(json.dumps({k: round(v, 1) for k, v in out.items()}))
