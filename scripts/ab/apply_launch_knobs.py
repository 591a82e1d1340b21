#!/usr/bin/env python3
"""A build of the library whose FAST apply launch rule reads two environment variables (never shipped):
  UHDR_X_MINBLK  the fewest blocks a launch may have before its cells per thread are halved (shipped: 448)
  UHDR_X_CPT     the most cells per thread (shipped: 32)
-> scripts/ab/libvar_K.so; scripts/ab/r4_single_knobs.sh sweeps them over the single-image calls."""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
TMP = "/tmp/uhdr_knobs/a/b"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function"]
SRCS = ["uhdr_kernels.hip", "uhdr_capi.hip", "uhdr_jpeg.hip", "uhdr_jpeg_dec.hip", "uhdr_jpeg_hdr.cpp", "uhdr_jpeg_prog.cpp", "uhdr_jpegr.cpp"]


def sub(s, old, new):
    assert old in s, old[:80]
    return s.replace(old, new, 1)


shutil.rmtree("/tmp/uhdr_knobs", ignore_errors=True)
os.makedirs(TMP)
shutil.copytree(os.path.join(ROOT, "libultrahdr_dev_amd", "csrc"), TMP + "/csrc")
shutil.copytree(os.path.join(ROOT, "include"), "/tmp/uhdr_knobs/a/include")
p = TMP + "/csrc/uhdr_kernels.hip"
s = open(p).read()
s = sub(s, '#include <cmath>\n', '#include <cmath>\n#include <cstdlib>\n')
s = sub(s, '''    uint32_t cpt = kApplyMaxCellsPerThread;''', '''    uint32_t cpt = getenv("UHDR_X_CPT") ? (uint32_t)atoi(getenv("UHDR_X_CPT")) : kApplyMaxCellsPerThread;
    const uint64_t minblk = getenv("UHDR_X_MINBLK") ? (uint64_t)atoi(getenv("UHDR_X_MINBLK")) : 448u;''')
s = sub(s, '''    while (cpt > 1u && blocks(cpt) < 448u) cpt >>= 1;''', '''    while (cpt > 1u && blocks(cpt) < minblk) cpt >>= 1;''')
open(p, "w").write(s)
out = os.path.join(ROOT, "scripts", "ab", "libvar_K.so")
subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-shared", "-o", out] + SRCS, cwd=TMP + "/csrc", stderr=subprocess.DEVNULL)
print("built", out)
