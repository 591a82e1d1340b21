#!/usr/bin/env python3
"""Throwaway builds of the library that take k_generate_resolve apart (round 3), each as scripts/ab/libvar_<name>.so:

  R1  the exact arithmetic removed (entries and pair loads kept)         -> what the f64 path costs
  R2  the loop body never runs                                            -> launch + counts + prefix sums + finalisation
  R3  every thread resolves one of its image's first 8 entries again      -> same instruction stream, every load a cache hit
  R4  counts instead of statistics: stat_out = (candidates, entries)      -> scripts/ab/count_lists.py prints them

Timed with  UHDR_HIP_LIB=$PWD/scripts/ab/libvar_R1.so bash scripts/kernel_times_of.sh scripts/time_step_loop.py
(64 x 4K, us per launch, one box):  shipped 26.5-27.3 | R1 23.1 | R2 6.1 | R3 13.1;  R4: 882 entries per image, 38 of them candidates.
The kernel is the latency of ONE round of 14 scattered line reads per pair (790 K requests per launch): 14 of its 27 us.
"""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function"]
SRCS = ["uhdr_kernels.hip", "uhdr_capi.hip", "uhdr_jpeg.hip", "uhdr_jpeg_dec.hip", "uhdr_jpeg_hdr.cpp", "uhdr_jpeg_prog.cpp", "uhdr_jpegr.cpp"]


def sub(s, old, new):
    assert s.count(old) == 1, (s.count(old), old[:70])
    return s.replace(old, new)


CALL = "    exact_pair<TF, ALIGNED, false>(c, im, im_v, my, pr, two, nullptr, nullptr, o, gn);\n"
LOADS = ("    { uint32_t hy[2][4][2], huv[2][2][2], y8[2][4], u8[2][2], v8[2][2];\n"
         "      load_pair<ALIGNED>(c, im, im_v, my, pr, two, hy, huv, y8, u8, v8);\n"
         "      uint32_t x = 0u;\n"
         "      for (int k = 0; k < 2; ++k) { for (int r = 0; r < 4; ++r) x ^= hy[k][r][0] ^ hy[k][r][1] ^ y8[k][r];\n"
         "        for (int r = 0; r < 2; ++r) x ^= huv[k][r][0] ^ huv[k][r][1] ^ u8[k][r] ^ v8[k][r]; }\n"
         "      o[0] = (uint8_t)x; o[1] = (uint8_t)(x >> 8); gn[0] = __uint_as_float(x & 0x3fffffffu); gn[1] = gn[0]; }\n")


def r4(s):
    s = sub(s, "    const uint32_t idx = entry >> 3;\n    const uint32_t my = idx / pairs_per_row, pr = idx - my * pairs_per_row;\n    const bool two = ALIGNED || (pr * 2u + 1u < c.map_w);\n    uint8_t o[2];",
            "    const uint32_t idx = entry >> 3;\n    if (entry & 4u) atomicAdd(&ws[6], 1u);\n    atomicAdd(&ws[7], 1u);\n    const uint32_t my = idx / pairs_per_row, pr = idx - my * pairs_per_row;\n    const bool two = ALIGNED || (pr * 2u + 1u < c.map_w);\n    uint8_t o[2];")
    return sub(s, "      c.stat_out[2u * img_i] = mn;\n      c.stat_out[2u * img_i + 1u] = mx;",
               "      c.stat_out[2u * img_i] = (float)__hip_atomic_load(&ws[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n"
               "      c.stat_out[2u * img_i + 1u] = (float)__hip_atomic_load(&ws[7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); (void)mn; (void)mx;")


VARIANTS = {
    "R1": lambda s: sub(s, CALL, LOADS),
    "R2": lambda s: sub(s, "  for (uint32_t g = blockIdx.y * 256u + threadIdx.x; g < n; g += kResolveSlices * 256u) {\n    uint32_t entry = (g << 3) | 7u;",
                        "  for (uint32_t g = blockIdx.y * 256u + threadIdx.x; g < n && c.height == 7u; g += kResolveSlices * 256u) {\n    uint32_t entry = (g << 3) | 7u;"),
    "R3": lambda s: sub(s, "      entry = ws[kStatHdr + l * kStatCap + (g - s_first[l])];", "      entry = ws[kStatHdr + 0u * kStatCap + (threadIdx.x & 7u)]; (void)l;"),
    "R4": r4,
}

if __name__ == "__main__":
    procs = []
    for name in (sys.argv[1:] or sorted(VARIANTS)):
        top = "/tmp/uhdr_v_%s" % name
        d = top + "/a/b"
        shutil.rmtree(top, ignore_errors=True)
        os.makedirs(d)
        shutil.copytree(ROOT + "/libultrahdr_dev_amd/csrc", d + "/csrc")
        shutil.copytree(ROOT + "/include", top + "/a/include")
        p = d + "/csrc/uhdr_kernels.hip"
        text = open(p).read()
        open(p, "w").write(VARIANTS[name](text))
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc"] + FLAGS + ["-shared", "-o", ROOT + "/scripts/ab/libvar_%s.so" % name] + SRCS,
                                      cwd=d + "/csrc", stderr=subprocess.DEVNULL))
    assert all(p.wait() == 0 for p in procs)
    print("built", len(procs))
