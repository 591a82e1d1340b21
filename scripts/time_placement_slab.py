#!/usr/bin/env python3
"""Placement inside ONE slab: the bench's batch (64 4K pairs: P010 arena, YUV arena, map arena, output arena) laid out at chosen byte
offsets of a single allocation; generate alone, apply alone and the step timed for every layout.
  python scripts/time_placement_slab.py base            # the whole batch shifted
  python scripts/time_placement_slab.py gaps            # gaps between the four arenas
  python scripts/time_placement_slab.py stride          # padding between frames of every arena
  python scripts/time_placement_slab.py slabs           # one packed batch per allocation, seven allocations"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from libultrahdr_dev_amd import api, synth, sharding
torch.cuda.set_device(0)
lib = api.init(0)
W, H, N = 3840, 2160, 64
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
fmt = api.OUTPUT_HDR_HLG
SLAB = torch.empty((14 << 30) if (len(sys.argv) < 2 or sys.argv[1] != "slabs") else (4700 << 20), dtype=torch.uint8, device="cuda")
SIZES = (W * H * 3, W * H * 3 // 2, (W // 4) * (H // 4), W * H * 4)   # p010, yuv, map, out
MiB = 1 << 20
SLABS = None


class Placed(bench.Batch):
    """bench.Batch with its four arenas at explicit offsets of the slab: starts[k] = byte offset of arena k, pads[k] = bytes between frames"""

    def __init__(self, starts, pads):
        self.lib, self.n, self.stats = lib, N, True
        ar = []
        for k, size in enumerate(SIZES):
            SLAB = SLABS[k] if SLABS is not None else globals()["SLAB"]   # (vmmarenas: every arena a slab of its own)
            if isinstance(starts[k], (list, tuple)):   # every frame's own offset
                assert all(o % 256 == 0 and o + size <= SLAB.numel() for o in starts[k])
                ar.append([SLAB[o:o + size] for o in starts[k]])
                continue
            stride = (size + 255) // 256 * 256 + pads[k]
            assert starts[k] % 256 == 0 and stride % 256 == 0 and starts[k] + stride * N <= SLAB.numel()
            ar.append([SLAB[starts[k] + i * stride:starts[k] + i * stride + size] for i in range(N)])
        self.p010, self.yuv, self.maps, self.outs = ar
        for i in range(N):
            synth.lcg_frame(W, H, sharding.image_seed(i), out=(self.p010[i], self.yuv[i]))
        self.minmax = torch.zeros(2 * N, dtype=torch.float32, device="cuda")
        self.yi = api.image_array([api.yuv420_image(y.data_ptr(), W, H, api.CG_BT709) for y in self.yuv])
        self.pi = api.image_array([api.p010_image(p.data_ptr(), W, H, api.CG_BT2100) for p in self.p010])
        self.mi = api.image_array([api.out_image(m.data_ptr()) for m in self.maps])
        self.oi = api.image_array([api.out_image(o.data_ptr()) for o in self.outs])
        self.md = api.Metadata()


def ms(f, n=30):
    for _ in range(4): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def step_ms(b, n=40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        b.generate(s); b.apply(s, fmt)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def packed(base, gaps=(0, 0, 0), pads=(0, 0, 0, 0)):
    starts, at = [], base
    for k, size in enumerate(SIZES):
        starts.append(at)
        at += ((size + 255) // 256 * 256 + pads[k]) * N + (gaps[k] if k < 3 else 0)
        at = (at + 255) // 256 * 256
    return starts, pads


def run(label, starts, pads):
    b = Placed(starts, pads)
    for _ in range(10): b.generate(s); b.apply(s, fmt)
    g, a, st = ms(lambda: b.generate(s)), ms(lambda: b.apply(s, fmt)), step_ms(b)
    print("%-44s generate %.4f  apply %.4f  step %.4f ms = %.0f MPix/s" % (label, g, a, st, N * W * H / st / 1e3), flush=True)


print("slab at %#x" % SLAB.data_ptr())
mode = sys.argv[1] if len(sys.argv) > 1 else "base"
if mode == "base":
    for base in (0, 4096, 65536, 1 * MiB, 2 * MiB, 4 * MiB, 8 * MiB, 16 * MiB, 32 * MiB, 64 * MiB, 128 * MiB, 256 * MiB, 512 * MiB, 1024 * MiB,
                 2048 * MiB, 3 * 1024 * MiB, 4096 * MiB, 6 * 1024 * MiB, 8 * 1024 * MiB, 0):
        run("batch at slab + %d KiB" % (base >> 10), *packed(base))
elif mode == "gaps":
    for g in (0, 4096, 65536, 1 * MiB, 2 * MiB, 6 * MiB, 16 * MiB, 50 * MiB, 128 * MiB, 250 * MiB, 512 * MiB, 1000 * MiB):
        run("gap %d KiB after every arena" % (g >> 10), *packed(0, (g, g, g)))
elif mode == "slabs":   # one batch per allocation, several allocations (all kept): is it the allocation?
    keep = [SLAB]
    for k in range(int(sys.argv[2]) if len(sys.argv) > 2 else 7):
        run("slab %d at %#x (4700 MiB)" % (k, SLAB.data_ptr()), *packed(0))
        SLAB = torch.empty(4700 << 20, dtype=torch.uint8, device="cuda")
        keep.append(SLAB)
elif mode in ("contig", "hipmalloc"):   # slabs straight from the HIP runtime: hipExtMallocWithFlags(hipDeviceMallocContiguous) / hipMalloc
    hiprt = C.CDLL("libamdhip64.so")
    hiprt.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    hiprt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]

    class Raw:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}

    keep = []
    for k in range(int(sys.argv[2]) if len(sys.argv) > 2 else 7):
        ptr, n = C.c_void_p(), 4700 << 20
        rc = hiprt.hipExtMallocWithFlags(C.byref(ptr), n, 0x4) if mode == "contig" else hiprt.hipMalloc(C.byref(ptr), n)
        if rc != 0:
            print("allocation %d failed: hipError %d" % (k, rc)); break
        raw = Raw(ptr.value, n)
        SLAB = torch.as_tensor(raw, device="cuda")
        keep.append((raw, SLAB))
        run("%s slab %d at %#x" % (mode, k, ptr.value), *packed(0))
elif mode == "scatter":   # a physically contiguous slab (the uniformly slow kind), frames at irregular offsets inside it
    import random
    hiprt = C.CDLL("libamdhip64.so")
    hiprt.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]

    class Raw:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}

    ptr, n = C.c_void_p(), 12 << 30
    assert hiprt.hipExtMallocWithFlags(C.byref(ptr), n, 0x4) == 0
    raw = Raw(ptr.value, n)
    SLAB = torch.as_tensor(raw, device="cuda")
    run("contiguous slab, packed", *packed(0))
    for gran in (4096, 65536, 2 * MiB):
        for trial in range(3):
            rnd = random.Random(gran + trial)
            starts, at = [], 0
            for k, size in enumerate(SIZES):
                room = 2 * ((size + gran - 1) // gran * gran)            # every frame owns a region twice its size, and sits somewhere in it
                starts.append([at + i * room + rnd.randrange(0, room - size + 1, gran) // 256 * 256 for i in range(N)])
                at += room * N
            run("frames at random offsets (multiples of %d KiB), draw %d" % (gran >> 10, trial), starts, (0, 0, 0, 0))
    for trial in range(3):   # the frames of all four arenas shuffled over one region
        rnd = random.Random(99 + trial)
        cell = (SIZES[3] + 2 * MiB - 1) // (2 * MiB) * (2 * MiB)
        order = list(range(4 * N))
        rnd.shuffle(order)
        starts = [[order[k * N + i] * cell for i in range(N)] for k in range(4)]
        run("all 256 frames shuffled over 2 MiB-aligned cells, draw %d" % trial, starts, (0, 0, 0, 0))
elif mode == "vmm":   # slabs whose physical backing is chosen piece by piece (scripts/ab/vmm_arena.cpp): chunk size, mapped in order or shuffled
    vmm = C.CDLL(os.path.join(ROOT, "scripts", "ab", "libvmm_arena.so"))
    vmm.vmm_alloc.argtypes = [C.c_size_t, C.c_size_t, C.c_uint, C.POINTER(C.c_void_p)]
    gran = C.c_size_t()
    print("vmm granularity rc=%d %d B" % (vmm.vmm_granularity(C.byref(gran)), gran.value))

    class Raw:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}

    keep = []
    run("torch slab (for this box's level)", *packed(0))
    for chunk, seed in ((2 * MiB, 0), (2 * MiB, 1), (2 * MiB, 2), (16 * MiB, 0), (16 * MiB, 1), (128 * MiB, 0), (128 * MiB, 1), (1024 * MiB, 0), (1024 * MiB, 1),
                        (2 * MiB, 3), (4700 * MiB, 0)):
        ptr, n = C.c_void_p(), 4700 << 20
        rc = vmm.vmm_alloc(n, chunk, seed, C.byref(ptr))
        if rc != 0:
            print("vmm_alloc(chunk %d MiB) failed: %d" % (chunk >> 20, rc)); continue
        raw = Raw(ptr.value, n)
        SLAB = torch.as_tensor(raw, device="cuda")
        keep.append((raw, SLAB))
        run("vmm slab, chunks of %d MiB, %s" % (chunk >> 20, "shuffled (seed %d)" % seed if seed else "mapped in order"), *packed(0))
elif mode == "vmmmap":   # many slabs of one kind, all kept: vmmmap CHUNK_MiB SEED COUNT
    vmm = C.CDLL(os.path.join(ROOT, "scripts", "ab", "libvmm_arena.so"))
    vmm.vmm_alloc.argtypes = [C.c_size_t, C.c_size_t, C.c_uint, C.POINTER(C.c_void_p)]

    class Raw:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}

    chunk, seed, count = int(sys.argv[2]) * MiB, int(sys.argv[3]), int(sys.argv[4])
    keep = []
    for k in range(count):
        ptr, n = C.c_void_p(), 4700 << 20
        rc = vmm.vmm_alloc(n, chunk, seed + k if seed else 0, C.byref(ptr))
        if rc != 0:
            print("vmm_alloc failed: %d" % rc); break
        raw = Raw(ptr.value, n)
        SLAB = torch.as_tensor(raw, device="cuda")
        keep.append((raw, SLAB))
        run("vmm slab %d, chunks of %d MiB%s" % (k, chunk >> 20, ", shuffled" if seed else ""), *packed(0))
elif mode == "cpt":   # UHDR_HIP_LIB=scripts/ab/libvar_K.so (scripts/ab/apply_launch_knobs.py): cells per thread of the FAST apply walk, contiguous slab
    hiprt = C.CDLL("libamdhip64.so")
    hiprt.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]

    class Raw:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}

    ptr, n = C.c_void_p(), 4700 << 20
    assert hiprt.hipExtMallocWithFlags(C.byref(ptr), n, 0x4) == 0
    raw = Raw(ptr.value, n)
    SLAB = torch.as_tensor(raw, device="cuda")
    for cpt in (32, 31, 33, 30, 28, 24, 20, 36, 40, 48, 27, 29, 32):
        os.environ["UHDR_X_CPT"] = str(cpt)
        run("physically contiguous slab, %d cells per thread" % cpt, *packed(0))
elif mode == "vmmperm":   # the SAME physical chunks under several mappings: vmmperm CHUNK_MiB SETS PERMS
    vmm = C.CDLL(os.path.join(ROOT, "scripts", "ab", "libvmm_arena.so"))
    vmm.vmm_create.argtypes, vmm.vmm_create.restype = [C.c_size_t, C.c_size_t], C.c_void_p
    vmm.vmm_map.argtypes, vmm.vmm_map.restype = [C.c_void_p, C.c_uint], C.c_void_p

    class Raw:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}

    chunk, sets, perms = int(sys.argv[2]) * MiB, int(sys.argv[3]), int(sys.argv[4])
    keep = []
    for k in range(sets):
        n = 4700 << 20
        ctx = vmm.vmm_create(n, chunk)
        assert ctx
        for seed in range(perms):
            torch.cuda.synchronize()
            va = vmm.vmm_map(ctx, seed)
            assert va
            raw = Raw(va, n)
            SLAB = torch.as_tensor(raw, device="cuda")
            run("chunk set %d (%d MiB chunks), %s" % (k, chunk >> 20, "permutation %d" % seed if seed else "in order"), *packed(0))
        keep.append(ctx)
elif mode == "vmmpool":   # a pool of chunks walking through device memory; slabs from consecutive chunks against slabs from every k-th: vmmpool CHUNK_MiB POOL_GiB
    vmm = C.CDLL(os.path.join(ROOT, "scripts", "ab", "libvmm_arena.so"))
    vmm.vmm_pool_create.argtypes, vmm.vmm_pool_create.restype = [C.c_size_t, C.c_size_t], C.c_void_p
    vmm.vmm_pool_chunks.argtypes, vmm.vmm_pool_chunks.restype = [C.c_void_p], C.c_size_t
    vmm.vmm_pool_map.argtypes, vmm.vmm_pool_map.restype = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t], C.c_void_p

    class Raw:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}

    chunk, total = int(sys.argv[2]) * MiB, int(sys.argv[3]) << 30
    pool = vmm.vmm_pool_create(total, chunk)
    nch = vmm.vmm_pool_chunks(pool)
    per = ((4700 << 20) + chunk - 1) // chunk
    slabs = nch // per
    print("pool of %d chunks of %d MiB; a slab takes %d: %d slabs" % (nch, chunk >> 20, per, slabs))
    keep = []
    for label, first, stride in [("consecutive chunks, slab %d" % k, k * per, 1) for k in range(0, slabs, max(1, slabs // 12))] + \
                                [("every %d-th chunk, from chunk %d" % (slabs, k), k, slabs) for k in range(0, slabs, max(1, slabs // 12))]:
        va = vmm.vmm_pool_map(pool, first, stride, per)
        assert va
        raw = Raw(va, per * chunk)
        SLAB = torch.as_tensor(raw, device="cuda")
        keep.append(raw)
        run(label, *packed(0))
elif mode == "vmmarenas":   # every arena a slab of its own from one pool of chunks: where each arena's chunks come from, separately
    vmm = C.CDLL(os.path.join(ROOT, "scripts", "ab", "libvmm_arena.so"))
    vmm.vmm_pool_create.argtypes, vmm.vmm_pool_create.restype = [C.c_size_t, C.c_size_t], C.c_void_p
    vmm.vmm_pool_chunks.argtypes, vmm.vmm_pool_chunks.restype = [C.c_void_p], C.c_size_t
    vmm.vmm_pool_map.argtypes, vmm.vmm_pool_map.restype = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t], C.c_void_p

    class Raw:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}

    chunk = 16 * MiB
    pool = vmm.vmm_pool_create(200 << 30, chunk)
    nch = vmm.vmm_pool_chunks(pool)
    need = [(((sz + 255) // 256 * 256) * N + chunk - 1) // chunk for sz in SIZES]   # chunks per arena: 100, 50, 3, 127
    print("pool of %d chunks; arenas take %s" % (nch, need))
    keep = []

    def arenas(label, spec):
        """spec[k] = (first chunk, stride) of arena k"""
        global SLABS
        SLABS = []
        for k in range(4):
            va = vmm.vmm_pool_map(pool, spec[k][0], spec[k][1], need[k])
            assert va
            raw = Raw(va, need[k] * chunk)
            keep.append(raw)
            SLABS.append(torch.as_tensor(raw, device="cuda"))
        run(label, [0, 0, 0, 0], (0, 0, 0, 0))

    q = nch // 4
    for rep in range(2):
        o = rep * 37
        arenas("all arenas: every k-th chunk of the whole pool", [(o + 0, nch // need[0]), (o + 1, nch // need[1]), (o + 2, nch // need[2]), (o + 3, nch // need[3])])
        arenas("every arena contiguous, a quarter of the pool apart", [(o + 0, 1), (o + q, 1), (o + 2 * q, 1), (o + 3 * q, 1)])
        arenas("inputs contiguous (far apart), output spread", [(o + 0, 1), (o + q, 1), (o + 2 * q, 1), (o + 3, nch // need[3])])
        arenas("inputs spread, output contiguous", [(o + 0, nch // need[0]), (o + 1, nch // need[1]), (o + 2, nch // need[2]), (o + 3 * q, 1)])
        arenas("every arena spread over its own quarter", [(o + 0, q // need[0]), (o + q, q // need[1]), (o + 2 * q, q // need[2]), (o + 3 * q, q // need[3])])
        arenas("P010 over the first half, YUV over the second, output over all", [(o + 0, (nch // 2) // need[0]), (o + nch // 2, (nch // 2) // need[1]), (o + 2, nch // need[2]), (o + 3, nch // need[3])])
        arenas("all arenas spread over the first 16 GiB", [(o + 0, 1024 // need[0]), (o + 1, 1024 // need[1]), (o + 2, 1024 // need[2]), (o + 3, 1024 // need[3])])
        arenas("all arenas spread over the first 64 GiB", [(o + 0, 4096 // need[0]), (o + 1, 4096 // need[1]), (o + 2, 4096 // need[2]), (o + 3, 4096 // need[3])])
elif mode == "stride":
    for p in (0, 256, 1024, 4096, 8192, 65536, 256 * 1024, 1 * MiB, 2 * MiB + 4096):
        run("%d B between frames (all arenas)" % p, *packed(0, (0, 0, 0), (p, p, p, p)))
