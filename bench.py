#!/usr/bin/env python3
"""Headline benchmark: MPixels/sec of gain-map generate + apply on an HBM-resident batch of
4K P010 + YUV420 pairs (BASELINE.json metric; workload = configs[2], "Batch 64 x 4K P010 frames,
generate+apply on 1 MI355X"; with N GPUs every rank holds its own 64 frames = configs[3]'s sharding).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Started without torch.distributed.run and with --gpus N > 1, this file launches the N ranks itself (fresh child processes through
torch.distributed.run, started before anything here has touched a GPU) and relays rank 0's JSON line.

One step = one pass of the hot path over the rank's batch: generateGainMap (HLG, P010 BT.2100 vs
SDR BT.709) into HBM-resident maps, then applyGainMap (FAST) of those maps -> RGBA1010102 HLG with
max_display_boost = FLT_MAX.  Inputs are already in HBM when the timed region starts; nothing crosses
PCIe inside it.  For N > 1 each step also all-reduces (RCCL) the batch's content min/max boost --
the only exchange the path has.  Rank 0 prints ONE JSON line.

MPix = width*height counted once per (generate+apply) pair.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

from libultrahdr_dev_amd import api, sharding, synth

W, H = 3840, 2160
CHUNK = 64                       # images per kernel launch (kMaxChunk in csrc/uhdr_kernels.h)
# algorithmic HBM bytes per 4K frame: every input byte read once, every output written once
GEN_BYTES = W * H * 3 + W * H * 3 // 2 + (W // 4) * (H // 4)           # 24 883 200 + 12 441 600 + 518 400
APP_BYTES = W * H * 3 // 2 + (W // 4) * (H // 4) + W * H * 4           # 12 441 600 + 518 400 + 33 177 600
GEN_READ_BYTES = W * H * 3 + W * H * 3 // 2                            # generate: P010 + YUV420 in
APP_READ_BYTES = W * H * 3 // 2 + (W // 4) * (H // 4)                  # apply: YUV420 + map in
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=64, help="4K frame pairs per step and GPU (one batch)")
    ap.add_argument("--rotate", type=int, default=3, help="resident batches per GPU: step k works on batch k mod R, so that no step "
                    "reads what the step before it touched (a fixed batch leaves part of its 8-bit chroma planes in the 256 MB Infinity "
                    "Cache for the next step's generate: ~2 %% of the rate, VERDICT r03).  `value` is this protocol; the fixed-batch "
                    "figure (R = 1) is reported beside it as fixed_batch")
    ap.add_argument("--ramp-ms", type=float, default=500.0, help="keep the GPU busy with the same step for this long before the W "
                    "warmup steps: the card raises its clocks only under sustained load (a 5-step warmup is 6 ms; measured 5 %% "
                    "between a cold and a ramped card).  Reported as clock_ramp_ms; 0 switches it off")
    ap.add_argument("--apply-format", default="hlg", choices=["hlg", "pq"])
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl == RCCL; gloo only to rehearse "
                    "the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--comm", default="torch", choices=["torch", "capi"], help="who issues the step's one exchange for N>1: torch.distributed "
                    "(default; the path the gloo rehearsals cover) or the C library (libuhdr_hip_comm.so: uhdr_hip_comm_allreduce_minmax, "
                    "the RCCL id carried to the ranks through torch.distributed's store).  capi needs backend nccl and one GPU per rank; "
                    "where it cannot be set up on EVERY rank the run falls back to torch and says why in collective.path_note")
    ap.add_argument("--exchange-with-one-rank", action="store_true", help="testing aid: with --gpus 1, still create the process group "
                    "(a world of one) and run the per-step exchange and the collective report -- the one way a one-GPU box can put "
                    "--comm capi's path through the bench on hardware.  Not the driver's protocol: the N = 1 headline has no exchange")
    ap.add_argument("--events", default="apply", choices=["all", "apply", "none"], help="which launches of the TIMED region are bracketed by HIP "
                    "events: the dominant kernel's (apply: what roofline.avg_launch_ms needs; default), both kernels', or none.  An event "
                    "record between two kernels keeps the second from starting while the first drains: bracketing both kernels costs the "
                    "step 1.6 %%, apply alone 0.6 %% (profiles/r04_bench_protocol.txt).  With `apply`, generate's launch time comes from a "
                    "pass of its own behind the timed region (both kernels bracketed there; kernels.generate says so)")
    ap.add_argument("--arena", default="spread", choices=["spread", "hipmalloc"], help="where the resident batches lie: `spread` takes them from "
                    "one placement pool of the library (uhdr_hip_mem_pool_*: every arena's physical chunks spaced evenly over the memory of all "
                    "resident batches), `hipmalloc` from torch's allocator (one hipMalloc per arena: physically contiguous on a free device).  "
                    "The other policy is measured behind `value` and reported beside it (`placement`).")
    ap.add_argument("--pool-gib", type=float, default=-1.0, help="--arena spread: the stretch of device memory the placement pool is drawn from, "
                    "in GiB.  The pool is created whole, the arenas take chunks spaced evenly over it, the rest goes back to the device "
                    "(uhdr_hip_mem_pool_trim) before anything is timed.  -1 (default) = 70 %% of the free device memory, at most 200 GiB: a pool "
                    "only as large as what stays resident (0) is fast or not by where it happens to lie, a wide one always "
                    "(profiles/r04_placement.txt)")
    ap.add_argument("--no-placement-ab", action="store_true", help="skip the measurement of the other placement policy (profiling runs)")
    ap.add_argument("--no-fixed-batch", action="store_true", help="skip the fixed-batch measurement behind `value` (profiling runs: the last "
                    "K dispatches of the process are then the K timed steps of `value`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the extra configs[1]/[4] timings (profiling runs)")
    ap.add_argument("--no-stats", action="store_true", help="experiment: generate without the per-image content min/max pass")
    ap.add_argument("--cpu-frames", type=int, default=3, help="frames of the same batch timed on the host CPU")
    ap.add_argument("--dry-run", action="store_true", help="launcher / rendezvous / reductions only, no GPU work and no measurement "
                    "(tests/test_bench_launcher.py: the N-rank launch on a box without GPUs); prints value null")
    return ap.parse_args()


def sources_sha16():
    """identifies the kernels a profile was taken with: profiles/traffic_latest.json carries it, and a line whose kernels have
    changed since reports traffic null instead of a stale number"""
    import hashlib
    h = hashlib.sha256()
    for f in ("uhdr_kernels.hip", "uhdr_kernels.h", "uhdr_device_math.h"):
        h.update(open(os.path.join(ROOT, "libultrahdr_dev_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def launch_ranks(a):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as children.  Nothing in this process has
    initialised a GPU (importing torch does not), and it stays that way: it only waits and passes the children's output on."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def workload_name(a, world, dry=False):
    """config.workload: BASELINE.json configs[2] on one GPU; weak scaling keeps 64 frames per GPU, which at 8 GPUs IS configs[3]
    (512 = 64 x 8 frames sharded 8-way, the min/max-boost all-reduce over RCCL)"""
    fmt = getattr(a, "apply_format", "hlg").upper()
    per = "%d x 3840x2160 P010(BT.2100,HLG)+YUV420(BT.709) per GPU, generate + apply(FAST)->RGBA1010102 %s, HBM-resident" % (a.frames, fmt)
    if world == 1:
        name = "configs[2]: batch " + per
    elif a.frames * world == 512 and world == 8:
        name = "configs[3]: batch 512 = 64 x 8 frames sharded 8-way (image i on rank i // 64), min/max-boost all-reduce over RCCL; " + per
    else:
        name = "configs[2] per GPU on %d GPUs (weak scaling towards configs[3] = 64 x 8): batch %d = " % (world, a.frames * world) + per
    return ("dry run, no GPU work: " if dry else "") + name


def collective_report(dist, backend, rank, world, device_label, local_mm, reduced, dev=None, iters=50, path="torch.distributed", exchange=None):
    """What lets a reader of rank 0's line check that the N ranks were real and that the path's one exchange crossed all of them:
    every rank's identity (host, pid, device), the reduced (min, max) checked against the ranks' own contributions gathered on the
    side, and the time of the 8-byte all-reduce by itself (median over `iters`, max over ranks).  Collective: every rank calls it."""
    import socket
    import statistics
    import torch
    ident = {"rank": rank, "host": socket.gethostname(), "pid": os.getpid(), "device": device_label}
    idents = [None] * world
    dist.all_gather_object(idents, ident)
    contrib = [None] * world
    dist.all_gather_object(contrib, [float(local_mm[0]), float(local_mm[1])])
    want = [min(c[0] for c in contrib), max(c[1] for c in contrib)]
    t = torch.zeros(2, dtype=torch.float32, device=dev if dev is not None else "cpu")
    times = []
    for _ in range(iters):
        if dev is not None:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        if exchange is not None:
            exchange()                     # (--comm capi: the C library's fold + ncclAllReduce + store, on the current stream)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if dev is not None:
            torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e6)
    med = torch.tensor([statistics.median(times)], dtype=torch.float64, device=dev if dev is not None else "cpu")
    dist.all_reduce(med, op=dist.ReduceOp.MAX)
    return {"backend": backend + (" (RCCL)" if backend == "nccl" else ""), "world": dist.get_world_size(), "ranks": idents,
            "path": path,
            "devices": [i["device"] for i in idents], "distinct_processes": len({(i["host"], i["pid"]) for i in idents}),
            "distinct_devices": len({(i["host"], i["device"]) for i in idents}),
            "allreduce_us": round(float(med.item()), 1), "allreduce_bytes": 8,
            "content_minmax": [float(reduced[0]), float(reduced[1])], "content_minmax_of_gathered_contributions": want,
            "reduction_checked": [float(reduced[0]), float(reduced[1])] == want}


def init_capi_comm(dist, backend, rank, world, dev):
    """--comm capi: one communicator of libuhdr_hip_comm.so per rank (include/uhdr_hip_comm.h), rank 0's RCCL id carried to the others
    through torch.distributed.  Collective: every rank calls it; every rank gets the same answer -- (lib, handle, None) or
    (None, None, reason) when ANY rank cannot take the path (the ranks agree before anything collective of RCCL's is called, so no
    rank waits in ncclCommInitRank for one that never comes).  Nothing is re-executed and no process is replaced."""
    def all_ok(ok):
        votes = [None] * world
        dist.all_gather_object(votes, bool(ok))
        return all(votes)

    lib, note = None, None
    if backend != "nccl":
        note = "--comm capi needs one GPU per rank (backend nccl); this run's backend is %s: the exchange stays on torch.distributed" % backend
    else:
        try:
            lib = api.load_comm()
        except (ImportError, OSError) as e:
            note = "libuhdr_hip_comm.so could not be loaded (%s): the exchange stays on torch.distributed" % e
    if not all_ok(note is None):
        return None, None, note or "another rank could not take the C library path: the exchange stays on torch.distributed"
    ident = (C.c_char * api.COMM_ID_BYTES)()
    box = [None]
    if rank == 0 and lib.uhdr_hip_comm_get_unique_id(ident) == 0:
        box[0] = bytes(ident.raw)
    dist.broadcast_object_list(box, src=0)
    if box[0] is None:
        return None, None, "uhdr_hip_comm_get_unique_id failed on rank 0: the exchange stays on torch.distributed"
    ident = (C.c_char * api.COMM_ID_BYTES).from_buffer_copy(box[0])
    handle = C.c_void_p()
    rc = lib.uhdr_hip_comm_init(ident, world, rank, dev, C.byref(handle))
    if not all_ok(rc == 0):
        if rc == 0:
            lib.uhdr_hip_comm_destroy(handle)
        return None, None, "uhdr_hip_comm_init failed on a rank (status %d on rank %d): the exchange stays on torch.distributed" % (rc, rank)
    return lib, handle, None


def dry_run(a, world, rank):
    """the multi-rank plumbing without a GPU: gloo rendezvous, the per-step content min/max reduction, the MAX over ranks of the
    elapsed time, one JSON line from rank 0"""
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    mm = torch.tensor([1.0 + rank, 4.0 + rank] * a.frames, dtype=torch.float32)
    red = torch.zeros(2, dtype=torch.float32)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        if world > 1:
            _, work = sharding.reduce_content_minmax(mm, dist, red, async_op=True)
            sharding.finish_content_minmax(red, work)
    elapsed = time.perf_counter() - t0
    coll = None
    if world > 1:
        assert dist.get_world_size() == a.gpus == world, (dist.get_world_size(), a.gpus, world)
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        note = None
        if getattr(a, "comm", "torch") == "capi":
            _, _, note = init_capi_comm(dist, "gloo", rank, world, 0)
        coll = collective_report(dist, "gloo", rank, world, "cpu", (1.0 + rank, 4.0 + rank), (float(red[0]), float(-red[1])))
        coll["path_note"] = note
    if rank == 0:
        print(json.dumps({"metric": "MPixels/sec gain-map generate+apply, 4K P010 batch", "value": None, "unit": "MPix/s", "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "dry_run": True, "scaling": "weak",
                          "content_minmax": None if world == 1 else [float(red[0]), float(-red[1])],
                          "collective": coll,
                          "config": {"workload": workload_name(a, world, dry=True), "frames_per_gpu": a.frames, "global_frames": a.frames * world}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


class Batch:
    """`frames` 4K pairs + their maps and 1010102 outputs, all resident in this rank's HBM."""

    ARENAS = (W * H * 3, W * H * 3 // 2, (W // 4) * (H // 4), W * H * 4)   # bytes per frame: P010, YUV, map, output

    @staticmethod
    def arena_bytes(size, frames):
        return (size + 255) // 256 * 256 * frames

    def __init__(self, lib, frames, rank, seed_offset=0, pool=None):
        self.lib, self.n = lib, frames
        self.stats = True
        self.p010, self.yuv, self.maps, self.outs = [], [], [], []
        # one arena per kind of buffer, frames back to back (256-byte multiples; measured: paddings between them change nothing --
        # what does is WHICH physical memory the arena occupies: `pool`)
        def arena(size, fill0):
            stride = (size + 255) // 256 * 256
            if pool is not None:
                t = pool.tensor(stride * frames)
                if fill0:
                    t.zero_()
            else:
                t = (torch.zeros if fill0 else torch.empty)(stride * frames, dtype=torch.uint8, device="cuda")
            return [t[i * stride:i * stride + size] for i in range(frames)]

        self.p010, self.yuv = arena(W * H * 3, False), arena(W * H * 3 // 2, False)
        self.maps, self.outs = arena((W // 4) * (H // 4), True), arena(W * H * 4, True)
        for i in range(frames):
            synth.lcg_frame(W, H, sharding.image_seed(seed_offset + rank * frames + i), out=(self.p010[i], self.yuv[i]))   # seed = 1234 + global image index
        self.minmax = torch.zeros(2 * frames, dtype=torch.float32, device="cuda")
        self.yi = api.image_array([api.yuv420_image(y.data_ptr(), W, H, api.CG_BT709) for y in self.yuv])
        self.pi = api.image_array([api.p010_image(p.data_ptr(), W, H, api.CG_BT2100) for p in self.p010])
        self.mi = api.image_array([api.out_image(m.data_ptr()) for m in self.maps])
        self.oi = api.image_array([api.out_image(o.data_ptr()) for o in self.outs])
        self.md = api.Metadata()

    def _slice(self, arr, lo):
        return C.cast(C.byref(arr, lo * C.sizeof(api.Image)), C.POINTER(api.Image))

    def generate(self, stream, events=None):
        for lo in range(0, self.n, CHUNK):
            m = min(CHUNK, self.n - lo)
            if events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            rc = self.lib.uhdr_hip_generate_gainmap_batch(
                m, self._slice(self.yi, lo), self._slice(self.pi, lo), api.TF_HLG, C.byref(self.md),
                self._slice(self.mi, lo), 0, C.c_void_p(self.minmax.data_ptr() + 8 * lo) if self.stats else None, stream)
            assert rc == 0, rc
            if events is not None:
                e1.record()
                events.append((e0, e1, m))

    def apply(self, stream, fmt, events=None):
        for lo in range(0, self.n, CHUNK):
            m = min(CHUNK, self.n - lo)
            if events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            rc = self.lib.uhdr_hip_apply_gainmap_batch(
                m, self._slice(self.yi, lo), self._slice(self.mi, lo), C.byref(self.md), fmt, api.FLT_MAX,
                self._slice(self.oi, lo), api.APPLY_FAST, stream)
            assert rc == 0, rc
            if events is not None:
                e1.record()
                events.append((e0, e1, m))


def other_configs(lib, stream):
    """BASELINE configs[1] and configs[4] timed outside the headline region (reported, not `value`):
    single 4K HLG generate latency, and 8K applyGainMap -> RGBA1010102 PQ / RGBA-F16 (max boost 10000/203)."""
    out = {}

    def timed(fn, iters=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    # configs[1]: one 3840x2160 pair, generate only
    p, y = synth.lcg_frame(W, H, 1234)
    m = torch.zeros((W // 4) * (H // 4), dtype=torch.uint8, device="cuda")
    yi, pi = api.yuv420_image(y.data_ptr(), W, H, api.CG_BT709), api.p010_image(p.data_ptr(), W, H, api.CG_BT2100)
    mi, md = api.out_image(m.data_ptr()), api.Metadata()
    ms = timed(lambda: lib.uhdr_hip_generate_gainmap(C.byref(yi), C.byref(pi), api.TF_HLG, C.byref(md), C.byref(mi), 0,
                                                     api.MEM_DEVICE, stream))
    out["configs[1] single 4K HLG generate"] = {"ms": round(ms, 4), "MPix/s": round(W * H / 1e6 / (ms * 1e-3), 1),
                                                "GB/s": round(GEN_BYTES / (ms * 1e-3) / 1e9, 1)}
    o1 = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda")
    mi1, oi1 = api.mono_image(m.data_ptr(), W // 4, H // 4), api.out_image(o1.data_ptr())
    ms = timed(lambda: lib.uhdr_hip_apply_gainmap(C.byref(yi), C.byref(mi1), C.byref(md), api.OUTPUT_HDR_HLG, api.FLT_MAX, C.byref(oi1),
                                                  api.APPLY_FAST, api.MEM_DEVICE, stream))
    out["single 4K apply -> HLG RGBA1010102"] = {"ms": round(ms, 4), "MPix/s": round(W * H / 1e6 / (ms * 1e-3), 1),
                                                 "GB/s": round(APP_BYTES / (ms * 1e-3) / 1e9, 1)}
    del o1
    # PQ generate (P010 BT.2100 PQ vs SDR BT.709): 2 f64 pow per HDR channel instead of 1 exp on the exact path, the f32 pre-filter
    # in front of it as for HLG -- 32 frames in one launch (8 left the chip half empty), plus the unfiltered kernel beside it
    nb = 32
    pq = [synth.lcg_frame(W, H, 4321 + i) for i in range(nb)]
    pmaps = [torch.zeros((W // 4) * (H // 4), dtype=torch.uint8, device="cuda") for _ in range(nb)]
    ya = api.image_array([api.yuv420_image(q[1].data_ptr(), W, H, api.CG_BT709) for q in pq])
    pa = api.image_array([api.p010_image(q[0].data_ptr(), W, H, api.CG_BT2100) for q in pq])
    ma = api.image_array([api.out_image(t.data_ptr()) for t in pmaps])
    ms = timed(lambda: lib.uhdr_hip_generate_gainmap_batch(nb, ya, pa, api.TF_PQ, C.byref(md), ma, 0, None, stream), 10)
    out["4K PQ generate, 32-frame launch"] = {"ms": round(ms, 4), "MPix/s": round(nb * W * H / 1e6 / (ms * 1e-3), 1),
                                              "GB/s": round(nb * GEN_BYTES / (ms * 1e-3) / 1e9, 1)}
    ms = timed(lambda: lib.uhdr_hip_generate_gainmap_batch_ex(nb, ya, pa, api.TF_PQ, C.byref(md), ma, 0, api.GENERATE_UNFILTERED, None, stream), 10)
    out["4K PQ generate, 32-frame launch, pre-filter off (UHDR_HIP_GENERATE_UNFILTERED)"] = {
        "ms": round(ms, 4), "MPix/s": round(nb * W * H / 1e6 / (ms * 1e-3), 1), "GB/s": round(nb * GEN_BYTES / (ms * 1e-3) / 1e9, 1)}
    # the two other pointwise loops of the encode path in their batched form (API-0's toneMap, API-0/1's convertYuv to BT.601):
    # 32 x 4K per call.  toneMap reads 3 B and writes 1.5 B per pixel; convertYuv reads and writes 1.5 B per pixel in place.
    tdst = [torch.zeros(W * H * 3 // 2, dtype=torch.uint8, device="cuda") for _ in range(nb)]
    ta = api.image_array([api.yuv420_image(t.data_ptr(), W, H, -1) for t in tdst])
    ms = timed(lambda: lib.uhdr_hip_tonemap_batch(nb, pa, ta, stream), 10)
    out["toneMap, 32 x 4K per call (uhdr_hip_tonemap_batch)"] = {"ms": round(ms, 4), "MPix/s": round(nb * W * H / 1e6 / (ms * 1e-3), 1),
                                                                "GB/s": round(nb * W * H * 4.5 / (ms * 1e-3) / 1e9, 1)}
    ms = timed(lambda: lib.uhdr_hip_tonemap(C.byref(pa[0]), C.byref(ta[0]), api.MEM_DEVICE, stream))
    out["toneMap, one 4K frame"] = {"ms": round(ms, 4), "GB/s": round(W * H * 4.5 / (ms * 1e-3) / 1e9, 1)}
    ms = timed(lambda: lib.uhdr_hip_convert_yuv_batch(nb, ta, api.CG_BT2100, api.CG_P3, stream), 10)
    out["convertYuv BT.2100 -> BT.601 in place, 32 x 4K per call (uhdr_hip_convert_yuv_batch)"] = {
        "ms": round(ms, 4), "MPix/s": round(nb * W * H / 1e6 / (ms * 1e-3), 1), "GB/s": round(nb * W * H * 3.0 / (ms * 1e-3) / 1e9, 1)}
    ms = timed(lambda: lib.uhdr_hip_convert_yuv(C.byref(ta[0]), api.CG_BT2100, api.CG_P3, api.MEM_DEVICE, stream))
    out["convertYuv, one 4K frame"] = {"ms": round(ms, 4), "GB/s": round(W * H * 3.0 / (ms * 1e-3) / 1e9, 1)}
    del tdst
    # opt-in LUT mode (upstream libultrahdr's USE_*_LUT configuration; bit-exact against the reference's LUT functions)
    ms = timed(lambda: lib.uhdr_hip_generate_gainmap_batch_ex(nb, ya, pa, api.TF_HLG, C.byref(md), ma, 0, api.GENERATE_LUT, None, stream), 10)
    out["LUT mode: 4K HLG generate, 32-frame launch"] = {"ms": round(ms, 4), "MPix/s": round(nb * W * H / 1e6 / (ms * 1e-3), 1),
                                                        "GB/s": round(nb * GEN_BYTES / (ms * 1e-3) / 1e9, 1)}
    louts = [torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda") for _ in range(nb)]
    mia = api.image_array([api.mono_image(t.data_ptr(), W // 4, H // 4) for t in pmaps])
    loa = api.image_array([api.out_image(t.data_ptr()) for t in louts])
    ms = timed(lambda: lib.uhdr_hip_apply_gainmap_batch(nb, ya, mia, C.byref(md), api.OUTPUT_HDR_HLG, api.FLT_MAX, loa, api.APPLY_LUT, stream), 10)
    out["LUT mode: 4K apply -> HLG RGBA1010102, 32-frame launch"] = {"ms": round(ms, 4), "MPix/s": round(nb * W * H / 1e6 / (ms * 1e-3), 1),
                                                                    "GB/s": round(nb * APP_BYTES / (ms * 1e-3) / 1e9, 1)}
    ms = timed(lambda: lib.uhdr_hip_apply_gainmap_batch(nb, ya, mia, C.byref(md), api.OUTPUT_HDR_HLG, api.FLT_MAX, loa, api.APPLY_EXACT, stream), 3)
    out["EXACT mode: 4K apply -> HLG RGBA1010102, 32-frame launch"] = {"ms": round(ms, 4), "MPix/s": round(nb * W * H / 1e6 / (ms * 1e-3), 1),
                                                                      "GB/s": round(nb * APP_BYTES / (ms * 1e-3) / 1e9, 1)}
    l8 = torch.zeros(W * H * 8, dtype=torch.uint8, device="cuda")
    l8a = api.image_array([api.out_image(l8.data_ptr())])
    for fmt, fname in ((api.OUTPUT_HDR_HLG, "HLG"), (api.OUTPUT_HDR_PQ, "PQ"), (api.OUTPUT_HDR_LINEAR, "F16"), (api.OUTPUT_HDR_LINEAR_RGB_10BIT, "planar 10-bit")):
        ms = timed(lambda: lib.uhdr_hip_apply_gainmap_batch(1, ya, mia, C.byref(md), fmt, api.FLT_MAX, l8a, api.APPLY_EXACT, stream), 10)
        out["EXACT mode: one 4K apply -> %s (the reference's bytes)" % fname] = {"ms": round(ms, 4), "MPix/s": round(W * H / 1e6 / (ms * 1e-3), 1)}
    del louts, l8
    # the drop-in form a CPU caller uses: host planes in, host bytes out (PCIe Gen5 both ways; never `value`)
    hp, hy = p.cpu().numpy(), y.cpu().numpy()
    hmap = np.zeros((W // 4) * (H // 4), np.uint8)
    hout = np.zeros(W * H, np.uint32)
    hyi = api.yuv420_image(hy.ctypes.data, W, H, api.CG_BT709)
    hpi = api.p010_image(hp.ctypes.data, W, H, api.CG_BT2100)
    hmi, hoi, hmd = api.out_image(hmap.ctypes.data), api.out_image(hout.ctypes.data), api.Metadata()

    def host_pair():
        assert lib.uhdr_hip_generate_gainmap(C.byref(hyi), C.byref(hpi), api.TF_HLG, C.byref(hmd), C.byref(hmi), 0, api.MEM_HOST, None) == 0
        assert lib.uhdr_hip_apply_gainmap(C.byref(hyi), C.byref(hmi), C.byref(hmd), api.OUTPUT_HDR_HLG, api.FLT_MAX, C.byref(hoi),
                                          api.APPLY_FAST, api.MEM_HOST, None) == 0

    for _ in range(3):
        host_pair()
    t0 = time.perf_counter()
    for _ in range(15):
        host_pair()
    ms = (time.perf_counter() - t0) / 15 * 1e3
    out["host-staged 4K pair (UHDR_HIP_MEM_HOST, pageable memory, PCIe both ways)"] = {
        "ms": round(ms, 3), "MPix/s": round(W * H / 1e6 / (ms * 1e-3), 1)}
    # editorhelper effects on one 4K YUV420 frame: byte gathers, read + write 12.4 MB each
    fx_out = torch.zeros(W * H * 3 // 2 + 64, dtype=torch.uint8, device="cuda")
    fin = api.Image(y.data_ptr(), W, H, api.CG_BT709, None, 0, 0, api.PIX_FMT_YUV420)
    fo = api.out_image(fx_out.data_ptr())
    for name, fn, fargs, obytes in (("rotate 90", lib.uhdr_hip_rotate, (90,), W * H * 3 // 2), ("mirror horizontal", lib.uhdr_hip_mirror, (1,), W * H * 3 // 2),
                                    ("resize to 1920x1080", lib.uhdr_hip_resize, (1920, 1080), 1920 * 1080 * 3 // 2)):
        ms = timed(lambda: fn(C.byref(fin), *fargs, C.byref(fo), api.MEM_DEVICE, stream), 10)
        out["4K YUV420 " + name] = {"ms": round(ms, 4), "GB/s (bytes written + bytes they come from)": round(2 * obytes / (ms * 1e-3) / 1e9, 1)}
    # SURVEY 8(f) rank 1, encode side: JpegEncoderHelper::compressImage on the device (FDCT + quantisation + Huffman + byte
    # stuffing), device planes in, device bytes out; the call ends with one stream synchronisation (it returns the size)
    sm = synth.smooth_frame(W, H, 77) if hasattr(synth, "smooth_frame") else None
    src = sm[1] if sm is not None else y
    jout = torch.zeros(W * H * 2, dtype=torch.uint8, device="cuda")
    jn = C.c_size_t()
    jimg = api.Image(src.data_ptr(), W, H, api.CG_BT709, src.data_ptr() + W * H, W, W // 2, api.PIX_FMT_YUV420)
    gimg = api.Image(m.data_ptr(), W // 4, H // 4, api.CG_UNSPECIFIED, None, W // 4, 0, api.PIX_FMT_MONOCHROME)
    for name, img, q, px in (("4K YUV420 q95", jimg, 95, W * H), ("960x540 gain map q85", gimg, 85, W * H // 16)):
        def enc():
            assert lib.uhdr_hip_jpeg_encode(C.byref(img), q, None, 0, C.c_void_p(jout.data_ptr()), jout.numel(), C.byref(jn), api.MEM_DEVICE, stream) == 0
        for _ in range(3):
            enc()
        t0 = time.perf_counter()
        for _ in range(10):
            enc()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        out["JPEG encode " + name + (" (smooth synthetic content)" if sm is not None else " (LCG noise)")] = {
            "ms": round(ms, 3), "MPix/s": round(px / 1e6 / (ms * 1e-3), 1), "jpeg_bytes": int(jn.value)}
    # ... and the decode side: JpegDecoderHelper::decompressImage(DECODE_TO_YCBCR) on the device (host JPEG bytes in, device planes out)
    assert lib.uhdr_hip_jpeg_encode(C.byref(jimg), 95, None, 0, C.c_void_p(jout.data_ptr()), jout.numel(), C.byref(jn), api.MEM_DEVICE, stream) == 0
    jbytes = jout[:int(jn.value)].cpu().numpy().copy()
    dplanes = torch.zeros(W * H * 3 // 2, dtype=torch.uint8, device="cuda")
    ddesc = api.Image()

    def dec():
        assert lib.uhdr_hip_jpeg_decode(C.c_void_p(jbytes.ctypes.data), jbytes.size, C.c_void_p(dplanes.data_ptr()), dplanes.numel(), C.byref(ddesc),
                                        api.MEM_DEVICE, stream) == 0
    for _ in range(3):
        dec()
    t0 = time.perf_counter()
    for _ in range(10):
        dec()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    out["JPEG decode 4K YUV420 q95 (the file encoded above; host bytes in, device planes out)"] = {
        "ms": round(ms, 3), "MPix/s": round(W * H / 1e6 / (ms * 1e-3), 1), "jpeg_bytes": int(jbytes.size)}
    # the whole codec calls -- encodeJPEGR API-1 / API-0 (device planes in, JPEG/R file on the host out) and
    # decodeJPEGR (host file in, device HDR rendition out), 4K and BASELINE configs[0]'s 640x480
    for cw, ch in ((W, H), (640, 480)):
        cp, cy = (sm if (cw, ch) == (W, H) and sm is not None else synth.smooth_frame(cw, ch, 78))
        cpi, cyi = api.p010_image(cp.data_ptr(), cw, ch, api.CG_BT2100), api.yuv420_image(cy.data_ptr(), cw, ch, api.CG_BT709)
        fbuf, fn_ = np.zeros(cw * ch * 3, np.uint8), C.c_size_t()

        def wall(fn, iters=10):
            for _ in range(3):
                assert fn() == 0
            t0 = time.perf_counter()
            for _ in range(iters):
                fn()
            return (time.perf_counter() - t0) / iters * 1e3
        t_api0 = wall(lambda: lib.uhdr_hip_jpegr_encode_api0(C.byref(cpi), api.TF_HLG, 95, None, 0, C.c_void_p(fbuf.ctypes.data), fbuf.size, C.byref(fn_),
                                                              api.MEM_DEVICE, stream))
        t_api1 = wall(lambda: lib.uhdr_hip_jpegr_encode_api1(C.byref(cpi), C.byref(cyi), api.TF_HLG, 95, None, 0, C.c_void_p(fbuf.ctypes.data), fbuf.size,
                                                              C.byref(fn_), api.MEM_DEVICE, stream))
        fbytes = fbuf[:fn_.value].copy()
        rend = torch.zeros(cw * ch * 4, dtype=torch.uint8, device="cuda")
        rdesc, rmd = api.Image(), api.Metadata()
        t_dec = wall(lambda: lib.uhdr_hip_jpegr_decode(C.c_void_p(fbytes.ctypes.data), fbytes.size, api.OUTPUT_HDR_HLG, api.FLT_MAX, C.c_void_p(rend.data_ptr()),
                                                       rend.numel(), C.byref(rdesc), C.byref(rmd), api.APPLY_FAST, api.MEM_DEVICE, stream))
        out["JPEG/R %dx%d HLG q95: encodeJPEGR API-0 / API-1, decodeJPEGR -> RGBA1010102 (wall clock per call)" % (cw, ch)] = {
            "encode_api0_ms": round(t_api0, 3), "encode_api1_ms": round(t_api1, 3), "decode_ms": round(t_dec, 3), "file_bytes": int(fbytes.size),
            "encode_api1_MPix/s": round(cw * ch / 1e6 / (t_api1 * 1e-3), 1), "decode_MPix/s": round(cw * ch / 1e6 / (t_dec * 1e-3), 1)}
    # ... and 16 files per call (uhdr_hip_jpegr_decode_batch: every decoder kernel covers all images of the call)
    nb16 = 16
    blobs = []
    for i in range(nb16):
        bp, by = synth.smooth_frame(W, H, 300 + i)
        bpi, byi = api.p010_image(bp.data_ptr(), W, H, api.CG_BT2100), api.yuv420_image(by.data_ptr(), W, H, api.CG_BT709)
        fb, fnn = np.zeros(W * H * 3, np.uint8), C.c_size_t()
        assert lib.uhdr_hip_jpegr_encode_api1(C.byref(bpi), C.byref(byi), api.TF_HLG, 95, None, 0, C.c_void_p(fb.ctypes.data), fb.size, C.byref(fnn), api.MEM_DEVICE, stream) == 0
        blobs.append(fb[:fnn.value].copy())
    del bp, by
    bptr = (C.c_void_p * nb16)(*[b.ctypes.data for b in blobs])
    bsz = (C.c_size_t * nb16)(*[b.size for b in blobs])
    bouts = [torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda") for _ in range(nb16)]
    boptr = (C.c_void_p * nb16)(*[o.data_ptr() for o in bouts])
    bocap = (C.c_size_t * nb16)(*[o.numel() for o in bouts])
    bdests, bmds, bstat = (api.Image * nb16)(), (api.Metadata * nb16)(), (C.c_int * nb16)()

    def dec16():
        assert lib.uhdr_hip_jpegr_decode_batch(nb16, bptr, bsz, api.OUTPUT_HDR_HLG, api.FLT_MAX, boptr, bocap, bdests, bmds, bstat, api.APPLY_FAST, api.MEM_DEVICE, stream) == 0
    for _ in range(3):
        dec16()
    t0 = time.perf_counter()
    for _ in range(5):
        dec16()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    out["JPEG/R decode, 16 x 4K files per call (uhdr_hip_jpegr_decode_batch) -> RGBA1010102"] = {
        "ms_per_call": round(ms, 3), "ms_per_file": round(ms / nb16, 3), "MPix/s": round(nb16 * W * H / 1e6 / (ms * 1e-3), 1)}
    del bouts
    # configs[4]: 7680x4320 decode-side apply
    w8, h8 = 7680, 4320
    _, y8 = synth.lcg_frame(w8, h8, 1234)
    m8 = torch.randint(0, 256, ((w8 // 4) * (h8 // 4),), dtype=torch.uint8, device="cuda")
    o8 = torch.zeros(w8 * h8 * 8, dtype=torch.uint8, device="cuda")
    yi8, mi8, oi8 = api.yuv420_image(y8.data_ptr(), w8, h8, api.CG_BT709), api.mono_image(m8.data_ptr(), w8 // 4, h8 // 4), api.out_image(o8.data_ptr())
    md8 = api.metadata(float(np.float32(10000.0) / np.float32(203.0)))
    for name, fmt, bpp in (("PQ RGBA1010102", api.OUTPUT_HDR_PQ, 4), ("linear RGBA-F16", api.OUTPUT_HDR_LINEAR, 8)):
        ms = timed(lambda: lib.uhdr_hip_apply_gainmap(C.byref(yi8), C.byref(mi8), C.byref(md8), fmt, api.FLT_MAX, C.byref(oi8),
                                                      api.APPLY_FAST, api.MEM_DEVICE, stream), 10)
        nbytes = w8 * h8 * 3 // 2 + (w8 // 4) * (h8 // 4) + w8 * h8 * bpp
        out["configs[4] 8K apply -> " + name] = {"ms": round(ms, 4), "MPix/s": round(w8 * h8 / 1e6 / (ms * 1e-3), 1),
                                                  "GB/s": round(nbytes / (ms * 1e-3) / 1e9, 1)}
    return out


def cpu_baseline(batch, fmt, nframes):
    """The oracle (kind "port": byte-identical to the reference on its golden vectors) timed on this
    box's host cores on the first `nframes` frames of rank 0's batch, and cross-checked against the GPU
    outputs of the same frames.  This is the ONLY use of oracle/ in this file."""
    from oracle import oracle as O
    ncpu = os.cpu_count() or 1
    torch.cuda.synchronize()
    t_gen = t_app = t_gen4 = t_app4 = 0.0
    worst, ndiff, nch = 0, 0, 0
    for i in range(nframes):
        p010 = batch.p010[i].cpu().numpy().view(np.uint16)
        yuv = batch.yuv[i].cpu().numpy()
        yi, pi = O.yuv420_image(yuv, W, H, O.CG_BT709), O.p010_image(p010, W, H, O.CG_BT2100)
        t0 = time.perf_counter()
        st, omap, omd = O.generate("orc_", yi, pi, O.TF_HLG, threads=ncpu)
        t1 = time.perf_counter()
        st2, ref, _ = O.apply("orc_", yi, omap, omd, fmt, api.FLT_MAX, threads=ncpu)
        t2 = time.perf_counter()
        assert st == 0 and st2 == 0
        t_gen += t1 - t0
        t_app += t2 - t1
        if i == 0:  # the reference's own threading policy: min(cores, 4) threads (ultrahdr.cpp:304,500)
            t0 = time.perf_counter()
            O.generate("orc_", yi, pi, O.TF_HLG, threads=0)
            t1 = time.perf_counter()
            O.apply("orc_", yi, omap, omd, fmt, api.FLT_MAX, threads=0)
            t2 = time.perf_counter()
            t_gen4, t_app4 = t1 - t0, t2 - t1
        gmap = batch.maps[i].cpu().numpy().reshape(omap.shape)
        assert np.array_equal(gmap, omap), "GPU gain map differs from the CPU oracle on frame %d" % i
        out = batch.outs[i].cpu().numpy().view(np.uint32)
        ref = ref.view(np.uint32)
        for sh in (0, 10, 20):
            d = np.abs(((out >> sh) & 0x3ff).astype(np.int32) - ((ref >> sh) & 0x3ff).astype(np.int32))
            worst = max(worst, int(d.max()))
            ndiff += int((d != 0).sum())
            nch += d.size
    mpix = W * H / 1e6
    # the next step of the reference's encode path on the CPU: libjpeg (the image's build behind the reference's call
    # sequence, one thread) on the smooth synthetic 4K frame other_configs times on the GPU; bytes cross-checked
    jpeg = None
    if O.load_libjpeg() is not None:
        _, sm = synth.smooth_frame(W, H, 77)
        hsm = sm.cpu().numpy()
        t0 = time.perf_counter()
        data = O.jpeg_encode("lj", hsm[:W * H], hsm[W * H:], W, H, 95)
        t_cpu = time.perf_counter() - t0
        jout = torch.zeros(W * H * 2, dtype=torch.uint8, device="cuda")
        jn = C.c_size_t()
        jimg = api.Image(sm.data_ptr(), W, H, api.CG_BT709, sm.data_ptr() + W * H, W, W // 2, api.PIX_FMT_YUV420)
        rc = batch.lib.uhdr_hip_jpeg_encode(C.byref(jimg), 95, None, 0, C.c_void_p(jout.data_ptr()), jout.numel(), C.byref(jn), api.MEM_DEVICE, None)
        same = rc == 0 and jn.value == len(data) and jout[:jn.value].cpu().numpy().tobytes() == data
        t0 = time.perf_counter()
        dst, dplanes, _, _, _ = O.jpeg_decode("lj", data)
        t_dec = time.perf_counter() - t0
        gp = torch.zeros(W * H * 3 // 2, dtype=torch.uint8, device="cuda")
        gd = api.Image()
        dbuf = np.frombuffer(data, np.uint8)
        rc = batch.lib.uhdr_hip_jpeg_decode(C.c_void_p(dbuf.ctypes.data), dbuf.size, C.c_void_p(gp.data_ptr()), gp.numel(), C.byref(gd), api.MEM_DEVICE, None)
        same_dec = rc == 0 and dst > 0 and np.array_equal(gp.cpu().numpy(), dplanes)
        jpeg = {"libjpeg_4k_yuv420_q95_ms": round(t_cpu * 1e3, 2), "MPix/s": round(mpix / t_cpu, 1), "bytes": len(data), "threads": 1,
                "gpu_bytes_identical": bool(same), "libjpeg_decode_ms": round(t_dec * 1e3, 2), "gpu_decoded_planes_identical": bool(same_dec)}
    # ... and the whole encodeJPEGR API-1 call on the CPU: the restatement of oracle/jpegr_oracle.py (gain map on all cores, convertYuv,
    # both JPEG compressions single-threaded as in the reference, container) on the same 4K frame pair; files cross-checked
    jpegr = None
    try:
        from oracle import jpegr_oracle as J
        smp, smy = synth.smooth_frame(W, H, 77)
        hp, hy = smp.cpu().numpy().view(np.uint16), smy.cpu().numpy()
        t0 = time.perf_counter()
        want = J.encode_api1(hp, hy, W, H, O.CG_BT709, O.CG_BT2100, O.TF_HLG, 95, threads=ncpu)
        t_enc = time.perf_counter() - t0
        fbuf, fn_ = np.zeros(W * H * 3, np.uint8), C.c_size_t()
        cpi, cyi = api.p010_image(smp.data_ptr(), W, H, api.CG_BT2100), api.yuv420_image(smy.data_ptr(), W, H, api.CG_BT709)
        rc = batch.lib.uhdr_hip_jpegr_encode_api1(C.byref(cpi), C.byref(cyi), api.TF_HLG, 95, None, 0, C.c_void_p(fbuf.ctypes.data), fbuf.size, C.byref(fn_),
                                                  api.MEM_DEVICE, None)
        t0 = time.perf_counter()
        dst, dref, _, _, _, _ = J.decode(want, O.OUT_HDR_HLG, api.FLT_MAX, threads=ncpu)
        t_dec = time.perf_counter() - t0
        rend = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda")
        rdesc, rmd = api.Image(), api.Metadata()
        wb = np.frombuffer(want, np.uint8)
        rc2 = batch.lib.uhdr_hip_jpegr_decode(C.c_void_p(wb.ctypes.data), wb.size, api.OUTPUT_HDR_HLG, api.FLT_MAX, C.c_void_p(rend.data_ptr()), rend.numel(),
                                              C.byref(rdesc), C.byref(rmd), api.APPLY_EXACT, api.MEM_DEVICE, None)
        jpegr = {"encode_api1_4k_cpu_ms": round(t_enc * 1e3, 1), "decode_4k_cpu_ms": round(t_dec * 1e3, 1), "file_bytes": len(want),
                 "gpu_file_identical": bool(rc == 0 and fbuf[:fn_.value].tobytes() == want),
                 "gpu_decoded_rendition_identical (EXACT mode)": bool(rc2 == 0 and dst == 0 and np.array_equal(rend.cpu().numpy(), dref))}
    except OSError:
        pass
    return {
        "jpeg_encode_cpu": jpeg, "jpegr_codec_cpu": jpegr,
        "value": round(nframes * mpix / (t_gen + t_app), 3), "unit": "MPix/s", "cores": ncpu, "kind": "port",
        "sample": "%d of the batch's 4K frames, generate+apply(%s), oracle/uhdr_oracle.c -O2 -ffp-contract=off, "
                  "%d row-band threads" % (nframes, "HLG" if fmt == api.OUTPUT_HDR_HLG else "PQ", ncpu),
        "generate_mpix_s": round(nframes * mpix / t_gen, 3), "apply_mpix_s": round(nframes * mpix / t_app, 3),
        "ref_policy_4_threads_mpix_s": round(mpix / (t_gen4 + t_app4), 3),
        # the reference's own gainmapmath.cpp object code (oracle/_ref) is timed in the container only and never loaded on the GPU
        # box: profiles/r02_reference_cpu.json (scripts/time_reference_cpu.py); it equals the oracle byte for byte there
        "gpu_vs_cpu_parity": {"map_bit_exact": True, "apply_worst_lsb": worst,
                              "apply_channels_differing": round(ndiff / max(nch, 1), 6)},
    }


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert a.gpus in (1, world), "--gpus %d but torch.distributed.run started %d ranks" % (a.gpus, world)
    if a.dry_run:
        return dry_run(a, world, rank)
    multi = world > 1 or a.exchange_with_one_rank
    dist = None
    if multi:
        import torch.distributed as dist
        if world == 1:   # (--exchange-with-one-rank outside a launcher: a rendezvous of one)
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(so.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path"
    ndev = torch.cuda.device_count()
    dev = local if local < ndev else local % ndev     # rehearsal on fewer GPUs than ranks shares devices (gloo only)
    assert a.backend != "nccl" or local < ndev, "RCCL needs one GPU per rank"
    torch.cuda.set_device(dev)
    if multi:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(a.backend)
        assert dist.get_world_size() == a.gpus == world, "ranks in the process group %d, --gpus %d, WORLD_SIZE %d" % (dist.get_world_size(), a.gpus, world)
    lib = api.init(dev)
    fmt = api.OUTPUT_HDR_HLG if a.apply_format == "hlg" else api.OUTPUT_HDR_PQ

    # R resident batches (R x 4.55 GB of the 288 GB): the timed steps rotate over them.  Batch 0 holds SURVEY 8(d)'s seeds
    # (1234 + global image index: what cpu_baseline cross-checks), the others the same generator 65536 seeds further on.
    R = max(1, a.rotate)
    pool, placement_note, pool_create_ms = None, None, None
    POOL_CHUNK = 16 << 20
    if a.arena == "spread":
        # one pool exactly the size of what stays resident; every arena is an allocation of its own, so its chunks are spaced evenly
        # over the pool: the R batches interleave physically
        need = sum((Batch.arena_bytes(sz, a.frames) + POOL_CHUNK - 1) // POOL_CHUNK * POOL_CHUNK for sz in Batch.ARENAS) * R
        try:
            span = a.pool_gib * (1 << 30) if a.pool_gib >= 0 else min(0.7 * torch.cuda.mem_get_info()[0], 200.0 * (1 << 30))
            t_pool = time.perf_counter()
            pool = api.MemPool(dev, max(need, int(span)), POOL_CHUNK)
            pool_create_ms = (time.perf_counter() - t_pool) * 1e3
        except Exception as e:   # (a runtime without the virtual-memory calls, a device without the memory: say so and go on)
            placement_note = "placement pool unavailable (%s): torch's allocator instead" % (str(e)[:200],)
    batches = [Batch(lib, a.frames, rank, seed_offset=65536 * r, pool=pool) for r in range(R)]
    pool_span_mib = None
    if pool is not None:
        pool_span_mib = pool.stats()[0] * (POOL_CHUNK >> 20)
        pool.trim()
    for bt in batches:
        bt.stats = not a.no_stats
    batch = batches[0]
    turn = [0]
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    red = torch.zeros(2, dtype=torch.float32, device="cuda")

    side = torch.cuda.Stream() if multi else None   # the exchange's own stream
    clib, chandle, comm_note = (None, None, None)
    if multi and a.comm == "capi":
        clib, chandle, comm_note = init_capi_comm(dist, a.backend, rank, world, dev)
    red_capi = torch.zeros(2, dtype=torch.float32, device="cuda")   # (min, max) as the C library leaves them

    def step(ev_gen=None, ev_app=None, exchange=True, rotate=True):
        batch = batches[turn[0] % R] if rotate else batches[0]
        turn[0] += 1
        batch.generate(stream, ev_gen)
        work = None
        if multi and exchange:
            # the path's only exchange: batch-wide content min / max boost (8 bytes, latency-bound).  Its small kernels (the fold of
            # this rank's pairs) and the all-reduce go onto a stream of their own behind generate, so that apply neither waits for
            # them nor has them in its way; the step ends when both streams have.
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            if clib is not None:
                rc = clib.uhdr_hip_comm_allreduce_minmax(chandle, C.c_void_p(batch.minmax.data_ptr()), batch.n, C.c_void_p(red_capi.data_ptr()),
                                                         C.c_void_p(side.cuda_stream))
                assert rc == 0, rc
            else:
                with torch.cuda.stream(side):
                    _, work = sharding.reduce_content_minmax(batch.minmax, dist, red, async_op=True)
        batch.apply(stream, fmt, ev_app)
        if multi and exchange:
            if work is not None:
                work.wait()                              # (RCCL: the current stream waits for the collective; gloo: the host does)
            torch.cuda.current_stream().wait_stream(side)

    def timed_steps(evs_g=None, evs_a=None, rotate=True):
        """W untimed steps, then exactly K timed ones between barriers + synchronisations; -> seconds (max over ranks)"""
        for _ in range(a.warmup):
            step(rotate=rotate)
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step(evs_g, evs_a, rotate=rotate)
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if multi:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # Setup takes a fraction of a second (the frames are written by one kernel each), so the card arrives here at idle clocks, and
    # it raises them only under sustained load.  Both states are measured with the same protocol and both are reported: first the
    # card as it comes (`cold_start`), then -- after the step has run for ramp_ms -- the state a service under load is in (`value`).
    cold_elapsed = timed_steps() if a.ramp_ms > 0 else None
    t_ramp = time.perf_counter()
    while (time.perf_counter() - t_ramp) * 1e3 < a.ramp_ms:
        for _ in range(8):
            step(exchange=False)   # (a time-bounded loop: ranks run different counts, so nothing collective in it)
        torch.cuda.synchronize()
    ev_gen, ev_app = [], []
    elapsed = timed_steps(ev_gen if a.events == "all" else None, ev_app if a.events != "none" else None)
    # the same protocol on ONE batch, step after step (rounds 1-3's `value`): reported beside `value`, never as it
    fixed_elapsed = timed_steps(rotate=False) if R > 1 and not a.no_fixed_batch else None
    gen_from_side_pass = False
    if a.events == "apply":   # generate's launch time: a pass of its own, both kernels bracketed (never part of `value`)
        side_gen, side_app = [], []
        timed_steps(side_gen, side_app)
        ev_gen, gen_from_side_pass = side_gen, True

    # the other placement policy, same protocol, right behind (never `value`): the batches again from torch's allocator
    other_elapsed = None
    if pool is not None and not a.no_placement_ab:
        mine = list(batches)
        batches[:] = [Batch(lib, a.frames, rank, seed_offset=65536 * r) for r in range(R)]
        for bt in batches:
            bt.stats = not a.no_stats
        other_elapsed = timed_steps()
        batches[:] = mine
        torch.cuda.empty_cache()

    def avg_ms_per_launch(evs):   # every event pair brackets exactly one kernel launch of <= CHUNK frames
        tot_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in evs)
        frames = sum(m for _, _, m in evs)
        return tot_ms / max(len(evs), 1), tot_ms, frames

    gen_ms, gen_tot, gen_frames = avg_ms_per_launch(ev_gen)
    app_ms, app_tot, app_frames = avg_ms_per_launch(ev_app)
    coll = None
    if multi:   # (outside the timed region; every rank takes part)
        mine = sharding.reduce_content_minmax(batch.minmax)             # this rank's own (min, max): no collective
        glob = sharding.reduce_content_minmax(batch.minmax, dist)       # the step's exchange once more, joined
        exch = None
        if clib is not None:   # ... through the C library when that is the path the steps took
            cur = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            exch = lambda: clib.uhdr_hip_comm_allreduce_minmax(chandle, C.c_void_p(batch.minmax.data_ptr()), batch.n, C.c_void_p(red_capi.data_ptr()), cur)
            assert exch() == 0
            torch.cuda.synchronize()
            glob = red_capi.clone()
        props = torch.cuda.get_device_properties(dev)
        label = "cuda:%d %s%s" % (dev, props.name, (" pci " + str(getattr(props, "pci_bus_id", ""))) if hasattr(props, "pci_bus_id") else "")
        coll = collective_report(dist, a.backend, rank, world, label, (float(mine[0]), float(mine[1])), (float(glob[0]), float(glob[1])),
                                 dev=torch.device("cuda", dev) if a.backend == "nccl" else None, exchange=exch,
                                 path="capi: libuhdr_hip_comm.so (uhdr_hip_comm_allreduce_minmax -> ncclAllReduce), id through torch.distributed" if clib is not None
                                 else "torch.distributed")
        coll["path_note"] = comm_note
        if clib is not None:
            clib.uhdr_hip_comm_destroy(chandle)

    if rank == 0:
        mpix_frame = W * H / 1e6
        total_frames = a.frames * world * a.steps
        value = total_frames * mpix_frame / elapsed
        gen_gbs = GEN_BYTES * gen_frames / (gen_tot * 1e-3) / 1e9 if gen_tot else 0.0   # (--events apply / none: not measured)
        app_gbs = APP_BYTES * app_frames / (app_tot * 1e-3) / 1e9 if app_tot else 0.0
        dominant = "apply" if app_tot >= gen_tot or a.events != "all" else "generate"
        ach = app_gbs if dominant == "apply" else gen_gbs
        per_launch = (APP_BYTES if dominant == "apply" else GEN_BYTES) * min(CHUNK, a.frames)
        traffic, traffic_src = None, None
        tp = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tp) and min(CHUNK, a.frames) == 64:   # HBM bytes per 64-frame launch from rocprofv3 --pmc passes of this same command
            try:
                tj = json.load(open(tp))
                if tj.get("sources_sha16") == sources_sha16():    # (scripts/profile_bench.sh stamps the kernels it profiled)
                    traffic, traffic_src = tj.get(dominant), tj.get("source")
                else:
                    traffic_src = "profiles/traffic_latest.json was taken with other kernel sources (stamp %s, now %s): not reported" % (
                        tj.get("sources_sha16"), sources_sha16())
            except Exception:
                pass
        # the box's own ceilings beside the 8 TB/s specification (torch kernels over 1 GiB buffers; < 1 s)
        ceilings = None
        try:
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            import mem_ceiling
            ceilings = {k: round(v, 1) for k, v in mem_ceiling.measure(1 << 30).items()}
        except Exception:
            pass
        read_gbs = (GEN_READ_BYTES + APP_READ_BYTES) * total_frames / world / elapsed / 1e9
        out = {
            "metric": "MPixels/sec gain-map generate+apply, 4K P010 batch", "value": round(value, 1), "unit": "MPix/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "clock_ramp_ms": a.ramp_ms, "ms_per_step": round(elapsed / a.steps * 1e3, 4),
            "cold_start": None if cold_elapsed is None else {
                "value": round(a.frames * world * a.steps * (W * H / 1e6) / cold_elapsed, 1), "ms_per_step": round(cold_elapsed / a.steps * 1e3, 4),
                "what": "the same W warmup + K timed steps run first, on the card as the setup leaves it (idle clocks); `value` is the same "
                        "measurement after the step has kept the card busy for clock_ramp_ms"},
            "protocol": {"resident_batches": R, "what": "step k works on batch k mod %d (each %d frames with inputs, maps and outputs of its own, all "
                         "resident in HBM before the timed region): no step reads or writes what the step before it touched" % (R, a.frames)},
            "fixed_batch": None if fixed_elapsed is None else {
                "value": round(a.frames * world * a.steps * (W * H / 1e6) / fixed_elapsed, 1), "ms_per_step": round(fixed_elapsed / a.steps * 1e3, 4),
                "what": "the same W + K steps over ONE batch (rounds 1-3's protocol), measured right after `value`: the next step's generate then "
                        "finds part of the chroma planes apply read in the 256 MB Infinity Cache"},
            "placement": {
                "policy": "spread" if pool is not None else "hipmalloc", "note": placement_note,
                "what": "which physical device memory the resident batches occupy (same bytes, same kernels, same virtual layout).  spread: "
                        "one uhdr_hip_mem_pool of %d MiB chunks drawn from pool_span_MiB of device memory, one allocation per arena of the %d "
                        "resident batches, each backed by chunks spaced evenly over the pool, the unused chunks returned before anything is timed "
                        "(pool_MiB stay); hipmalloc: one hipMalloc per arena (torch's allocator), physically contiguous on a device whose memory is "
                        "free.  DESIGN.md 6.1, profiles/r04_placement.txt" % (POOL_CHUNK >> 20, R),
                "pool_MiB": None if pool is None else pool.stats()[0] * (POOL_CHUNK >> 20), "pool_span_MiB": pool_span_mib,
                "pool_create_ms": None if pool_create_ms is None else round(pool_create_ms, 1),
                "hipmalloc": None if other_elapsed is None else {
                    "value": round(a.frames * world * a.steps * (W * H / 1e6) / other_elapsed, 1), "ms_per_step": round(other_elapsed / a.steps * 1e3, 4),
                    "what": "the same W + K rotating steps over batches from torch's allocator, measured right behind `value` in this process"}},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/u16 in, f32+f64 math, u8/u32 out",
            "data": "synthetic",
            "config": {"workload": workload_name(a, world),
                       "frames_per_gpu": a.frames, "global_frames": a.frames * world, "resident_batches_per_gpu": R, "width": W, "height": H,
                       "images_per_launch": min(CHUNK, a.frames),
                       "parallelism": "one image batch per GPU, no pixel traffic between GPUs"},
            "roofline": {"bound": "hbm", "kernel": "k_apply_s4<HLG>" if dominant == "apply" else "k_generate<HLG,aligned>",
                         "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                         "frac_note": "HIP events on the launch stream around each launch of the dominant kernel inside the timed region; consecutive "
                                      "kernels overlap at their boundaries where no event record sits between them, so per-kernel times do not "
                                      "add up to the step (profiles/r03_trace_vs_events.txt, profiles/r04_bench_protocol.txt); frac_whole_step "
                                      "is the number to quote",
                         "events_in_timed_region": a.events,
                         "frac_whole_step": round((GEN_BYTES + APP_BYTES) * total_frames / world / elapsed / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": per_launch,
                         "avg_launch_ms": round(app_ms if dominant == "apply" else gen_ms, 4),
                         # north_star's "HBM-read roofline": the bytes the step READS (50 284 800 per frame pair) over the step time
                         "read_only": {"achieved": round(read_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(read_gbs / HBM_PEAK_GBS, 4),
                                       "bytes_per_frame_pair": GEN_READ_BYTES + APP_READ_BYTES, "scope": "generate+apply step, per GPU"},
                         "measured_ceilings_this_box_GBs": ceilings},
            "kernels": {
                "generate": {"avg_launch_ms": round(gen_ms, 4), "GB/s": round(gen_gbs, 1), "frac_of_8TBs": round(gen_gbs / HBM_PEAK_GBS, 4),
                             "bytes_per_frame": GEN_BYTES,
                             "measured_in": "a pass of its own behind the timed region, both kernels bracketed by events" if gen_from_side_pass
                                            else "the timed region"},
                "apply": {"avg_launch_ms": round(app_ms, 4), "GB/s": round(app_gbs, 1), "frac_of_8TBs": round(app_gbs / HBM_PEAK_GBS, 4),
                          "bytes_per_frame": APP_BYTES},
                "generate+apply_GB/s": round((GEN_BYTES + APP_BYTES) * total_frames / world / elapsed / 1e9, 1),
                "frac_of_8TBs_whole_step_per_gpu": round((GEN_BYTES + APP_BYTES) * total_frames / world / elapsed / 1e9 / HBM_PEAK_GBS, 4),
            },
        }
        if coll is not None:
            out["collective"] = coll
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(batch, fmt, max(1, min(a.cpu_frames, a.frames)))
        if world == 1 and not a.no_other_configs:
            out["other_configs"] = other_configs(lib, stream)
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
