"""GPU: include/uhdr_hip_comm.h on the one GPU there is -- a world of one rank (RCCL accepts it): the fold of this rank's per-image
(min, max) pairs, the all-reduce and the store, on the caller's stream, against numpy; the (min, max) the generate call wrote go
through unchanged.  More ranks need more GPUs (RCCL wants a device per rank): the N-rank form of the same exchange is covered over
gloo by tests/test_sharding_gloo.py and tests/test_bench_launcher.py, and measured only by the driver's SCALE run."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_world_of_one_reduces_this_ranks_images():
    import torch
    from libultrahdr_dev_amd import api
    api.init(0)
    lib = api.load_comm()
    ident = (C.c_char * api.COMM_ID_BYTES)()
    assert lib.uhdr_hip_comm_get_unique_id(ident) == 0
    comm = C.c_void_p()
    assert lib.uhdr_hip_comm_init(ident, 1, 0, 0, C.byref(comm)) == 0 and comm.value
    w, r = C.c_int(-1), C.c_int(-1)
    assert lib.uhdr_hip_comm_world(comm, C.byref(w), C.byref(r)) == 0 and (w.value, r.value) == (1, 0)
    rng = np.random.default_rng(5)
    stream = torch.cuda.Stream()
    for images in (1, 3, 64, 1000):
        mm = rng.uniform(0.01, 50.0, (images, 2)).astype(np.float32)
        mm.sort(axis=1)
        d = torch.from_numpy(mm.reshape(-1).copy()).cuda()
        out = torch.full((2,), 7.0, dtype=torch.float32, device="cuda")
        with torch.cuda.stream(stream):
            assert lib.uhdr_hip_comm_allreduce_minmax(comm, C.c_void_p(d.data_ptr()), images, C.c_void_p(out.data_ptr()), C.c_void_p(stream.cuda_stream)) == 0
        stream.synchronize()
        got = out.cpu().numpy()
        assert got[0] == mm[:, 0].min() and got[1] == mm[:, 1].max(), (images, got)
    # the result may not lie inside the input: batch_minmax is the reduction buffer from the fold on (no scratch of the
    # communicator's that two exchanges in flight could meet in)
    d = torch.from_numpy(np.arange(8, dtype=np.float32)).cuda()
    assert lib.uhdr_hip_comm_allreduce_minmax(comm, C.c_void_p(d.data_ptr()), 4, C.c_void_p(d.data_ptr() + 8), None) == api.ERROR_BAD_PTR
    # ... and two exchanges on two streams at once, each with its own result, do not disturb each other
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    a, b2 = torch.tensor([3., 9., 2., 5.], device="cuda"), torch.tensor([7., 8., 6., 11.], device="cuda")
    ra, rb = torch.zeros(2, device="cuda"), torch.zeros(2, device="cuda")
    torch.cuda.synchronize()
    for _ in range(20):
        assert lib.uhdr_hip_comm_allreduce_minmax(comm, C.c_void_p(a.data_ptr()), 2, C.c_void_p(ra.data_ptr()), C.c_void_p(s1.cuda_stream)) == 0
        assert lib.uhdr_hip_comm_allreduce_minmax(comm, C.c_void_p(b2.data_ptr()), 2, C.c_void_p(rb.data_ptr()), C.c_void_p(s2.cuda_stream)) == 0
    torch.cuda.synchronize()
    assert ra.tolist() == [2.0, 9.0] and rb.tolist() == [6.0, 11.0]
    # a rank without images contributes nothing: with no image anywhere the result is (+inf, -inf)
    out = torch.zeros(2, dtype=torch.float32, device="cuda")
    assert lib.uhdr_hip_comm_allreduce_minmax(comm, None, 0, C.c_void_p(out.data_ptr()), None) == 0
    torch.cuda.synchronize()
    assert out.cpu().numpy().tolist() == [float("inf"), float("-inf")]
    assert lib.uhdr_hip_comm_destroy(comm) == 0


def test_generate_statistics_go_through_the_exchange():
    """the (min, max) pairs uhdr_hip_generate_gainmap_batch writes are what is reduced: four 640x480 LCG pairs"""
    import torch
    from libultrahdr_dev_amd import api, synth
    lib = api.init(0)
    cl = api.load_comm()
    w, h, n = 640, 480, 4
    frames = [synth.lcg_frame(w, h, 1234 + i) for i in range(n)]
    maps = [torch.zeros((w // 4) * (h // 4), dtype=torch.uint8, device="cuda") for _ in range(n)]
    yi = api.image_array([api.yuv420_image(f[1].data_ptr(), w, h, api.CG_BT709) for f in frames])
    pi = api.image_array([api.p010_image(f[0].data_ptr(), w, h, api.CG_BT2100) for f in frames])
    mi = api.image_array([api.out_image(m.data_ptr()) for m in maps])
    md = api.Metadata()
    mm = torch.zeros(2 * n, dtype=torch.float32, device="cuda")
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.uhdr_hip_generate_gainmap_batch(n, yi, pi, api.TF_HLG, C.byref(md), mi, 0, C.c_void_p(mm.data_ptr()), s) == 0
    ident = (C.c_char * api.COMM_ID_BYTES)()
    comm = C.c_void_p()
    assert cl.uhdr_hip_comm_get_unique_id(ident) == 0 and cl.uhdr_hip_comm_init(ident, 1, 0, 0, C.byref(comm)) == 0
    out = torch.zeros(2, dtype=torch.float32, device="cuda")
    assert cl.uhdr_hip_comm_allreduce_minmax(comm, C.c_void_p(mm.data_ptr()), n, C.c_void_p(out.data_ptr()), s) == 0
    torch.cuda.synchronize()
    v = mm.cpu().numpy().reshape(n, 2)
    assert out.cpu().numpy().tolist() == [float(v[:, 0].min()), float(v[:, 1].max())] and v[:, 0].min() < v[:, 1].max()
    assert cl.uhdr_hip_comm_destroy(comm) == 0


def _build_example(tmp_path):
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "multi_gpu_batch")
    libdir = os.path.join(root, "libultrahdr_dev_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(root, "include"),
                           "-o", exe, os.path.join(root, "examples", "multi_gpu_batch.cpp"), "-L" + libdir, "-luhdr_hip", "-luhdr_hip_comm",
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_host_program_one_rank(tmp_path):
    """examples/multi_gpu_batch.cpp -- the sharded step from C++ alone (one process per GPU, no Python in the ranks) -- with the one
    rank a one-GPU box allows: it runs, and the batch statistics it reports are those of the same four frames through the bindings"""
    import json
    import subprocess
    import torch
    from libultrahdr_dev_amd import api, synth
    exe = _build_example(tmp_path)
    r = subprocess.run([exe, "--gpus", "1", "--frames", "4", "--steps", "3", "--warmup", "1"], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["ranks_agree"] is True and d["value"] > 0
    lib = api.init(0)
    w, h, n = 3840, 2160, 4
    frames = [synth.lcg_frame(w, h, 1234 + i) for i in range(n)]
    maps = [torch.zeros((w // 4) * (h // 4), dtype=torch.uint8, device="cuda") for _ in range(n)]
    yi = api.image_array([api.yuv420_image(f[1].data_ptr(), w, h, api.CG_BT709) for f in frames])
    pi = api.image_array([api.p010_image(f[0].data_ptr(), w, h, api.CG_BT2100) for f in frames])
    mi = api.image_array([api.out_image(m.data_ptr()) for m in maps])
    md = api.Metadata()
    mm = torch.zeros(2 * n, dtype=torch.float32, device="cuda")
    assert lib.uhdr_hip_generate_gainmap_batch(n, yi, pi, api.TF_HLG, C.byref(md), mi, 0, C.c_void_p(mm.data_ptr()), None) == 0
    torch.cuda.synchronize()
    v = mm.cpu().numpy().reshape(n, 2)
    # (printed with nine significant digits: enough to name a float exactly)
    assert [np.float32(x) for x in d["content_minmax"]] == [v[:, 0].min(), v[:, 1].max()]


def test_bench_takes_the_c_library_path_with_comm_capi():
    """bench.py --comm capi on the one rank a one-GPU box allows (--exchange-with-one-rank: a process group and an RCCL communicator
    of one): the RCCL id travels through torch.distributed, every step's exchange is uhdr_hip_comm_allreduce_minmax on the side
    stream, and the line says which path ran and that the reduced pair equals the rank's own contribution."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--frames", "4", "--steps", "3", "--warmup", "1", "--ramp-ms", "0", "--rotate", "2",
                        "--exchange-with-one-rank", "--comm", "capi", "--no-cpu-baseline", "--no-other-configs"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    c = d["collective"]
    assert c["path"].startswith("capi") and c["path_note"] is None and c["world"] == 1 and c["backend"].startswith("nccl")
    assert c["reduction_checked"] is True and c["content_minmax"][0] < c["content_minmax"][1] and c["allreduce_us"] > 0
    assert d["n_gpus"] == 1 and d["value"] > 0
    # (the resident batches came from a placement pool -- the default --, and the line carries the other policy beside it)
    pl = d["placement"]
    assert pl["policy"] == "spread" and pl["note"] is None and pl["pool_MiB"] >= 2 * 4 * 71 and pl["hipmalloc"]["value"] > 0
