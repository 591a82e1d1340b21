"""world_size-2 CPU test (gloo) of the N>1 path: batch sharding + the content min/max all-reduce.
The per-image statistics come from the oracle (test infrastructure) on small frames; the reduction code
is the one bench.py runs over RCCL on the GPUs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_the_batch():
    from libultrahdr_dev_amd import sharding
    for n in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = sharding.shard_range(n, r, world)
                assert 0 <= lo <= hi <= n and (hi - lo) - n // world in (0, 1)
                seen += list(range(lo, hi))
            assert seen == list(range(n))
    assert sharding.shard_range(512, 3, 8) == (192, 256)        # BASELINE configs[3]: 64 frames per GPU
    assert sharding.image_seed(5) == 1239


def _worker(rank, world, port, n_images, q):
    sys.path.insert(0, ROOT)
    from libultrahdr_dev_amd import sharding
    from oracle import oracle as O
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharding.shard_range(n_images, rank, world)
    w, h = 64, 32
    stats = []
    for i in range(lo, hi):
        p010, yuv = O.lcg_frame(w, h, sharding.image_seed(i))
        st, m, md, (mn, mx) = O.generate("orc_", O.yuv420_image(yuv, w, h, 0), O.p010_image(p010, w, h, 2), 1, stats=True)
        assert st == 0
        stats += [mn, mx]
    t = torch.tensor(stats, dtype=torch.float32)
    g = sharding.reduce_content_minmax(t, dist)
    q.put((rank, lo, hi, float(g[0]), float(g[1])))
    dist.barrier()
    dist.destroy_process_group()


def test_minmax_allreduce_two_ranks_gloo(orc):
    from libultrahdr_dev_amd import sharding
    n_images, world = 7, 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_images, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # serial truth
    mins, maxs = [], []
    for i in range(n_images):
        p010, yuv = orc.lcg_frame(64, 32, sharding.image_seed(i))
        _, _, _, (mn, mx) = orc.generate("orc_", orc.yuv420_image(yuv, 64, 32, 0), orc.p010_image(p010, 64, 32, 2), 1, stats=True)
        mins.append(np.float32(mn)); maxs.append(np.float32(mx))
    covered = []
    for rank, lo, hi, gmin, gmax in res:
        assert np.float32(gmin) == min(mins) and np.float32(gmax) == max(maxs)
        covered += list(range(lo, hi))
    assert sorted(covered) == list(range(n_images))


def test_minmax_single_process_and_empty():
    from libultrahdr_dev_amd import sharding
    t = torch.tensor([1.0, 4.0, 0.5, 3.0, 2.0, 9.0])
    g = sharding.reduce_content_minmax(t)
    assert g.tolist() == [0.5, 9.0]
    g = sharding.reduce_content_minmax(torch.empty(0))
    assert g[0] == float("inf") and g[1] == float("-inf")
