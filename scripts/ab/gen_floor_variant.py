#!/usr/bin/env python3
"""k_generate's memory floor inside the library and the bench step (scripts/ab/libvar_GE1.so; never shipped, not the reference's
bytes): the kernel's fourteen loads per pixel pair, one XOR over them, its 2-byte store; no sampling, no transfer functions, no
statistics, no lists (k_generate_resolve still runs behind it and finds nothing).  VARIANTS="B GE1" bash scripts/ab/run_ab.sh"""
import os, shutil, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function"]
SRCS = ["uhdr_kernels.hip", "uhdr_capi.hip", "uhdr_jpeg.hip", "uhdr_jpeg_dec.hip", "uhdr_jpeg_hdr.cpp", "uhdr_jpeg_prog.cpp", "uhdr_jpegr.cpp"]
shutil.rmtree("/tmp/uhdr_ge1", ignore_errors=True)
d = "/tmp/uhdr_ge1/a/b"
os.makedirs(d)
shutil.copytree(os.path.join(ROOT, "libultrahdr_dev_amd", "csrc"), d + "/csrc")
shutil.copytree(os.path.join(ROOT, "include"), "/tmp/uhdr_ge1/a/include")
p = d + "/csrc/uhdr_kernels.hip"
s = open(p).read()
old = "    const uint32_t ex = gen_pair<TF, LUT, FILTER, DEFER>(c, hy, huv, y8, u8, v8, o, gn, s_srgb, s_hdr, (FILTER && DEFER) ? &mid : nullptr);"
assert s.count(old) == 1
new = """    uint32_t ex = 3u;
    if (FILTER && DEFER) {
      uint32_t x = 0u;
      for (int k = 0; k < 2; ++k) { for (int r = 0; r < 4; ++r) x ^= hy[k][r][0] ^ hy[k][r][1] ^ y8[k][r]; for (int r = 0; r < 2; ++r) x ^= huv[k][r][0] ^ huv[k][r][1] ^ u8[k][r] ^ v8[k][r]; }
      o[0] = (uint8_t)x; o[1] = (uint8_t)(x >> 8); gn[0] = gn[1] = 1.0f;
    } else ex = gen_pair<TF, LUT, FILTER, DEFER>(c, hy, huv, y8, u8, v8, o, gn, s_srgb, s_hdr, nullptr);"""
open(p, "w").write(s.replace(old, new))
subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-shared", "-o", os.path.join(ROOT, "scripts", "ab", "libvar_GE1.so")] + SRCS, cwd=d + "/csrc", stderr=subprocess.DEVNULL)
print("built GE1")
