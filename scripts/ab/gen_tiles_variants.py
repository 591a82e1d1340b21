#!/usr/bin/env python3
"""k_generate with 1 / 2 / 8 spans per block instead of 4 (scripts/ab/libvar_T<k>.so; never shipped): round 4 found one-shot blocks
to be the lower memory floor for apply's bytes -- does generate, a pure read stream, care?  VARIANTS="B T1 T2 T8" run_ab.sh"""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function"]
SRCS = ["uhdr_kernels.hip", "uhdr_capi.hip", "uhdr_jpeg.hip", "uhdr_jpeg_dec.hip", "uhdr_jpeg_hdr.cpp", "uhdr_jpeg_prog.cpp", "uhdr_jpegr.cpp"]
procs = []
shutil.rmtree("/tmp/uhdr_tiles", ignore_errors=True)
for k in (1, 2, 8):
    d = "/tmp/uhdr_tiles/%d/b" % k
    os.makedirs(d)
    shutil.copytree(os.path.join(ROOT, "libultrahdr_dev_amd", "csrc"), d + "/csrc")
    shutil.copytree(os.path.join(ROOT, "include"), "/tmp/uhdr_tiles/%d/include" % k)
    p = d + "/csrc/uhdr_kernels.hip"
    s = open(p).read()
    old = "constexpr int kGenBlock = 256, kGenTiles = 4;"
    assert old in s
    open(p, "w").write(s.replace(old, "constexpr int kGenBlock = 256, kGenTiles = %d;" % k))
    procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc"] + FLAGS + ["-shared", "-o", os.path.join(ROOT, "scripts", "ab", "libvar_T%d.so" % k)] + SRCS,
                                  cwd=d + "/csrc", stderr=subprocess.DEVNULL))
for pr in procs:
    assert pr.wait() == 0
print("built T1 T2 T8")
