#!/usr/bin/env python3
"""FAST apply with only G of the launch's images in flight at a time (shipped: all of them, block b works on image b mod n): scripts/ab/libvar_G<G>.so
for G = 8, 16, 32 (never shipped).  Round 2 had found all 64 together 5 % faster than image by image -- on hipMalloc memory; asked again on pool
memory.  VARIANTS="B G8 G16 G32" bash scripts/ab/run_ab.sh"""
import os, shutil, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function"]
SRCS = ["uhdr_kernels.hip", "uhdr_capi.hip", "uhdr_jpeg.hip", "uhdr_jpeg_dec.hip", "uhdr_jpeg_hdr.cpp", "uhdr_jpeg_prog.cpp", "uhdr_jpegr.cpp"]
procs = []
shutil.rmtree("/tmp/uhdr_group", ignore_errors=True)
old = """  const uint32_t img_i = blockIdx.x, span = blockIdx.y;
  const AppImage& im = b.img[img_i];
  const uint32_t idx = span * c.cells_per_thread * kApplyBlock + threadIdx.x;"""
for g in (8, 16, 32):
    d = "/tmp/uhdr_group/%d/b" % g
    os.makedirs(d)
    shutil.copytree(os.path.join(ROOT, "libultrahdr_dev_amd", "csrc"), d + "/csrc")
    shutil.copytree(os.path.join(ROOT, "include"), "/tmp/uhdr_group/%d/include" % g)
    p = d + "/csrc/uhdr_kernels.hip"
    s = open(p).read()
    assert s.count(old) == 1
    new = """  // (variant: groups of G images; a launch whose image count G does not divide keeps the shipped order)
  uint32_t img_i = blockIdx.x, span = blockIdx.y;
  if (gridDim.x %% %du == 0u) {
    const uint32_t L = blockIdx.y * gridDim.x + blockIdx.x, per_group = %du * gridDim.y, grp = L / per_group, r = L - grp * per_group;
    img_i = grp * %du + r %% %du;
    span = r / %du;
  }
  const AppImage& im = b.img[img_i];
  const uint32_t idx = span * c.cells_per_thread * kApplyBlock + threadIdx.x;""" % (g, g, g, g, g)
    open(p, "w").write(s.replace(old, new))
    procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc"] + FLAGS + ["-shared", "-o", os.path.join(ROOT, "scripts", "ab", "libvar_G%d.so" % g)] + SRCS,
                                  cwd=d + "/csrc", stderr=subprocess.DEVNULL))
for pr in procs:
    assert pr.wait() == 0
print("built G8 G16 G32")
