"""-m gpu: the placement pools of include/uhdr_hip.h ("where resident images lie in device memory"; no reference counterpart): device
memory taken as chunks through the HIP virtual-memory calls, every allocation one range of virtual addresses backed by chunks spaced
evenly over the pool's free ones.  What is tested is the allocator -- sizes, statuses, no aliasing, memory returned -- and that the
pixel path gives the oracle's bytes on images that live in such memory; what placement does to speed is measured, not tested
(scripts/time_placement_slab.py, profiles/r04_placement.txt)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
FLT_MAX = 3.4028234663852886e38
MiB = 1 << 20


def test_pool_allocations_sizes_statuses_and_memory_returned(hip):
    lib = hip.load()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()                               # (torch's own cache out of the free-memory readings below)
    free0 = torch.cuda.mem_get_info()[0]
    pool = hip.MemPool(0, 1000 * MiB)                      # rounded up to 63 chunks of 16 MiB
    assert pool.stats() == (63, 63)
    assert free0 - torch.cuda.mem_get_info()[0] >= 63 * 16 * MiB
    a = pool.tensor(100 * MiB)                             # 7 chunks
    b = pool.tensor(33 * MiB)                              # 3 chunks
    assert pool.stats() == (63, 53)
    a.fill_(0x5A)
    b.fill_(0xC3)
    torch.cuda.synchronize()
    assert int(a.min()) == int(a.max()) == 0x5A and int(b.min()) == int(b.max()) == 0xC3, "allocations alias"
    ramp = torch.arange(a.numel() // 4, dtype=torch.int32, device="cuda")
    a.view(torch.int32).copy_(ramp)                        # (every byte of the range is mapped, in order)
    assert torch.equal(a.view(torch.int32), ramp) and int(b.min()) == 0xC3
    p = C.c_void_p()
    assert lib.uhdr_hip_mem_pool_alloc(pool.handle, 54 * 16 * MiB, C.byref(p)) == hip.ERROR_INSUFFICIENT_RESOURCE and not p.value
    assert lib.uhdr_hip_mem_pool_alloc(pool.handle, 0, C.byref(p)) == hip.ERROR_UNSUPPORTED_FEATURE
    assert lib.uhdr_hip_mem_pool_alloc(pool.handle, 16, None) == hip.ERROR_BAD_PTR
    assert lib.uhdr_hip_mem_pool_free(pool.handle, C.c_void_p(a.data_ptr() + 4096)) == hip.ERROR_UNSUPPORTED_FEATURE
    aptr = a.data_ptr()
    del a, ramp
    torch.cuda.synchronize()
    assert pool.free(aptr) == 0 and pool.stats() == (63, 60)
    c = pool.tensor(60 * 16 * MiB)                         # everything that is left, exactly
    assert pool.stats() == (63, 0)
    c[::4096].fill_(7)
    assert int(b.max()) == 0xC3
    cptr = c.data_ptr()
    del c
    torch.cuda.synchronize()
    assert pool.free(cptr) == 0
    torch.cuda.empty_cache()
    before_trim = torch.cuda.mem_get_info()[0]
    assert pool.trim() == 0 and pool.stats() == (3, 0)
    assert torch.cuda.mem_get_info()[0] - before_trim >= 59 * 16 * MiB, "trim did not return the unused chunks to the device"
    assert lib.uhdr_hip_mem_pool_alloc(pool.handle, 16 * MiB, C.byref(p)) == hip.ERROR_INSUFFICIENT_RESOURCE
    assert int(b.min()) == int(b.max()) == 0xC3            # what is allocated survives a trim
    del b
    assert pool.destroy() == 0
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    assert free0 - torch.cuda.mem_get_info()[0] < 64 * MiB, "destroy did not return the pool's memory"   # (the kernels torch loaded meanwhile take some)
    # argument checks of create
    h = C.c_void_p()
    assert lib.uhdr_hip_mem_pool_create(0, 64 * MiB, 3 * MiB, C.byref(h)) == hip.ERROR_UNSUPPORTED_FEATURE
    assert lib.uhdr_hip_mem_pool_create(0, 0, 0, C.byref(h)) == hip.ERROR_UNSUPPORTED_FEATURE
    assert lib.uhdr_hip_mem_pool_create(99, 64 * MiB, 0, C.byref(h)) == hip.ERROR_UNSUPPORTED_FEATURE
    assert lib.uhdr_hip_mem_pool_create(0, 64 * MiB, 0, None) == hip.ERROR_BAD_PTR
    assert lib.uhdr_hip_mem_pool_create(0, 1 << 50, 0, C.byref(h)) == hip.ERROR_INSUFFICIENT_RESOURCE and not h.value
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info()[0] < 64 * MiB, "a pool that could not be created kept memory"


def test_pixel_path_on_images_that_live_in_a_pool(hip, orc):
    """generate (batched, with statistics) -> EXACT apply -> FAST apply on frames, maps and renditions that all live in pool memory,
    2 MiB chunks so that every image straddles several: the oracle's bytes."""
    from tests.gpu_util import diff_1010102
    lib = hip.load()
    n, w, h = 40, 640, 360
    pool = hip.MemPool(0, 400 * MiB, 2 * MiB)
    host, yis, pis, mis, ois, keep = [], [], [], [], [], []
    for i in range(n):
        p010, yuv = orc.lcg_frame(w, h, 4100 + i)
        dp, dy = pool.tensor(p010.nbytes), pool.tensor(yuv.nbytes)
        dp.copy_(torch.from_numpy(p010.view(np.uint8)))
        dy.copy_(torch.from_numpy(yuv))
        dm, do = pool.tensor((w // 4) * (h // 4)), pool.tensor(w * h * 4)
        keep += [dp, dy, dm, do]
        yis.append(hip.yuv420_image(dy.data_ptr(), w, h, hip.CG_BT709))
        pis.append(hip.p010_image(dp.data_ptr(), w, h, hip.CG_BT2100))
        mis.append(hip.out_image(dm.data_ptr()))
        ois.append(hip.out_image(do.data_ptr()))
        host.append((p010, yuv, dm, do))
    ya, pa, ma, oa = hip.image_array(yis), hip.image_array(pis), hip.image_array(mis), hip.image_array(ois)
    md = hip.Metadata()
    mm = torch.zeros(2 * n, dtype=torch.float32, device="cuda")
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.uhdr_hip_generate_gainmap_batch(n, ya, pa, hip.TF_HLG, C.byref(md), ma, 0, C.c_void_p(mm.data_ptr()), s) == 0
    assert lib.uhdr_hip_apply_gainmap_batch(n, ya, ma, C.byref(md), hip.OUTPUT_HDR_HLG, FLT_MAX, oa, hip.APPLY_EXACT, s) == 0
    torch.cuda.synchronize()
    refs = []
    for p010, yuv, dm, do in host[:8]:
        st, omap, omd = orc.generate("orc_", orc.yuv420_image(yuv, w, h, 0), orc.p010_image(p010, w, h, 2), 1)
        assert np.array_equal(dm.cpu().numpy().reshape(omap.shape), omap)
        st, ref, _ = orc.apply("orc_", orc.yuv420_image(yuv, w, h, 0), omap, omd, orc.OUT_HDR_HLG, FLT_MAX)
        assert np.array_equal(do.cpu().numpy(), ref.view(np.uint8).reshape(-1)), "EXACT apply on pool memory differs from the oracle"
        refs.append(ref)
    assert lib.uhdr_hip_apply_gainmap_batch(n, ya, ma, C.byref(md), hip.OUTPUT_HDR_HLG, FLT_MAX, oa, hip.APPLY_FAST, s) == 0
    torch.cuda.synchronize()
    for (p010, yuv, dm, do), ref in zip(host[:8], refs):
        worst, frac, ok = diff_1010102(do.cpu().numpy().view(np.uint32), ref.view(np.uint32))
        assert ok and worst <= 1
    del keep, host, dp, dy, dm, do
    assert pool.destroy() == 0
