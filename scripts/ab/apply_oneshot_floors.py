#!/usr/bin/env python3
"""Is a ONE-SHOT block (every thread one map cell, then the wave ends) a lower memory floor for k_apply_s4 than the 32-cell walk,
inside the library and the bench step?  scripts/ab/linear_apply_floor.hip says so stand-alone (0.45-0.51 against 0.49-0.55 ms per
64 x 4K, by box); scripts/ab/apply_shortlived_floors.py's XN1 could not tell, because the walk's straight-line form issues the
"next cell's" loads even when there is none -- with one cell per thread every load twice.  Builds (loads, one XOR, stores; never
shipped, not the reference's bytes):

  XP32   the walk, shipped prologue (37 KiB), 32 cells per thread                    (= X1 of apply_floor_variants.py)
  XO0    one-shot: the cell's ten loads, no table prologue, four stores, end        (512-thread blocks)
  XO13   one-shot + a 13 KiB prologue (stage 1 + a stage-2 table replicated 8 x instead of 32 x would be that size)
  XO37   one-shot + the shipped 37 KiB prologue

    python scripts/ab/apply_oneshot_floors.py && VARIANTS="XP32 XO0 XO13 XO37" bash scripts/ab/run_ab.sh
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
TMP = "/tmp/uhdr_oneshot/a/b"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function"]
SRCS = ["uhdr_kernels.hip", "uhdr_capi.hip", "uhdr_jpeg.hip", "uhdr_jpeg_dec.hip", "uhdr_jpeg_hdr.cpp", "uhdr_jpeg_prog.cpp", "uhdr_jpegr.cpp"]
# name -> (one-shot, prologue bytes: -1 = shipped)
MODES = {"XP32": (0, -1), "XO0": (1, 0), "XO13": (1, 13 * 1024), "XO37": (1, -1)}


def sub(s, old, new):
    assert old in s, old[:80]
    return s.replace(old, new, 1)


def main():
    want = sys.argv[1:] or list(MODES)
    shutil.rmtree("/tmp/uhdr_oneshot", ignore_errors=True)
    os.makedirs(TMP)
    shutil.copytree(os.path.join(ROOT, "libultrahdr_dev_amd", "csrc"), TMP + "/csrc")
    shutil.copytree(os.path.join(ROOT, "include"), "/tmp/uhdr_oneshot/a/include")
    p = TMP + "/csrc/uhdr_kernels.hip"
    s = open(p).read()
    s = sub(s, '#include "uhdr_kernels.h"\n', '#include "uhdr_kernels.h"\n#ifndef UHDR_ONESHOT\n#define UHDR_ONESHOT 0\n#endif\n#ifndef UHDR_PRO\n#define UHDR_PRO -1\n#endif\n')
    s = sub(s, "constexpr uint32_t kApplyMaxCellsPerThread = 32;", "constexpr uint32_t kApplyMaxCellsPerThread = UHDR_ONESHOT ? 1 : 32;")
    # the walk's floor (X1)
    s = sub(s, '''  if (T::kOetf) {
    if (interior) apply_cell_piped<FMT>(''', '''  {
    const uint32_t x = cur.yrow[0] ^ cur.yrow[1] ^ cur.yrow[2] ^ cur.yrow[3] ^ cur.uu[0] ^ cur.uu[1] ^ cur.vv[0] ^ cur.vv[1] ^ __float_as_uint(e1 + e2 + e3 + e4);
    for (int oy = 0; oy < 4; ++oy) st_stream(reinterpret_cast<uint4*>(static_cast<char*>(im.dst) + ((4u * cy + oy) * c.width + 4u * cx) * 4u), make_uint4(x, x + oy, x ^ 1u, x ^ 2u));
    cx = ncx; cy = ncy;
    return more;
  }
  if (T::kOetf) {
    if (interior) apply_cell_piped<FMT>(''')
    # prologue size
    s = sub(s, '''    constexpr uint32_t kN1 = T::kS1Bytes / 16u, kPer1 = (kN1 + kApplyBlock - 1u) / kApplyBlock;
    constexpr uint32_t kN2 = T::kOetf ? kTabS2Cells * 32u : 0u, kPer2 = (kN2 + kApplyBlock - 1u) / kApplyBlock;''',
            '''    constexpr uint32_t kN1 = UHDR_PRO == 0 ? 1u : T::kS1Bytes / 16u, kPer1 = (kN1 + kApplyBlock - 1u) / kApplyBlock;
    constexpr uint32_t kN2 = UHDR_PRO == 0 ? 0u : UHDR_PRO > 0 ? (UHDR_PRO - T::kS1Bytes) / 8u : T::kOetf ? kTabS2Cells * 32u : 0u, kPer2 = (kN2 + kApplyBlock - 1u) / kApplyBlock;''')
    # one-shot: the cell whose inputs were requested before the prologue, its stores, the end
    s = sub(s, '''  uint32_t left = c.cells_per_thread;
#pragma unroll 1
  for (;;) {''', '''#if UHDR_ONESHOT
  {
    const uint32_t x = ca.yrow[0] ^ ca.yrow[1] ^ ca.yrow[2] ^ ca.yrow[3] ^ ca.uu[0] ^ ca.uu[1] ^ ca.vv[0] ^ ca.vv[1] ^ ca.mrow[0] ^ ca.mrow[1] ^ (uint32_t)lut[(ca.yrow[0] * 2654435761u >> 20) & 0xff0u];
    for (int oy = 0; oy < 4; ++oy) st_stream(reinterpret_cast<uint4*>(static_cast<char*>(im.dst) + ((4u * cy + oy) * c.width + 4u * cx) * 4u), make_uint4(x, x + oy, x ^ 1u, x ^ 2u));
    return;
  }
#endif
  uint32_t left = c.cells_per_thread;
#pragma unroll 1
  for (;;) {''')
    open(p, "w").write(s)
    procs = []
    for name in want:
        one, pro = MODES[name]
        out = os.path.join(ROOT, "scripts", "ab", "libvar_%s.so" % name)
        procs.append((name, subprocess.Popen(["/opt/rocm/bin/hipcc"] + FLAGS + ["-DUHDR_ONESHOT=%d" % one, "-DUHDR_PRO=%d" % pro, "-shared", "-o", out] + SRCS,
                                             cwd=TMP + "/csrc", stderr=subprocess.PIPE)))
    for name, pr in procs:
        err = pr.communicate()[1].decode()
        assert pr.returncode == 0, (name, [l for l in err.splitlines() if "error" in l][:5])
    print("built", " ".join("scripts/ab/libvar_%s.so" % n for n in want))


if __name__ == "__main__":
    main()
