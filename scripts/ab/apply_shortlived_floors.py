#!/usr/bin/env python3
"""VERDICT r03 item 2, the floor first: would a SHORT-LIVED block of k_apply_s4 (1-8 cells per thread instead of 32, which only a
small table prologue allows) move its bytes faster?  Builds of the kernel's loads and stores with one XOR in between (X1 of
apply_floor_variants.py) and

  XN<k>  NO table prologue at all (the best any smaller table could do) and k cells per thread, k = 1, 2, 4, 8, 32
  XP<k>  the shipped 37 KiB prologue and k cells per thread (XP32 == X1)

(never shipped; not the reference's bytes).    python scripts/ab/apply_shortlived_floors.py && VARIANTS="XP32 XN32 XN8 XN4 XN2 XN1 XP4" bash scripts/ab/run_ab.sh
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
TMP = "/tmp/uhdr_short/a/b"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function"]
SRCS = ["uhdr_kernels.hip", "uhdr_capi.hip", "uhdr_jpeg.hip", "uhdr_jpeg_dec.hip", "uhdr_jpeg_hdr.cpp", "uhdr_jpeg_prog.cpp", "uhdr_jpegr.cpp"]


def sub(s, old, new):
    assert old in s, old[:80]
    return s.replace(old, new, 1)


def main():
    want = sys.argv[1:] or ["XP32", "XN32", "XN8", "XN4", "XN2", "XN1", "XP4"]
    shutil.rmtree("/tmp/uhdr_short", ignore_errors=True)
    os.makedirs(TMP)
    shutil.copytree(os.path.join(ROOT, "libultrahdr_dev_amd", "csrc"), TMP + "/csrc")
    shutil.copytree(os.path.join(ROOT, "include"), "/tmp/uhdr_short/a/include")
    p = TMP + "/csrc/uhdr_kernels.hip"
    s = open(p).read()
    s = sub(s, '#include "uhdr_kernels.h"\n', '#include "uhdr_kernels.h"\n#ifndef UHDR_NOPRO\n#define UHDR_NOPRO 0\n#endif\n#ifndef UHDR_CPT\n#define UHDR_CPT 32\n#endif\n')
    s = sub(s, "constexpr uint32_t kApplyMaxCellsPerThread = 32;", "constexpr uint32_t kApplyMaxCellsPerThread = UHDR_CPT;")
    s = sub(s, '''  if (T::kOetf) {
    if (interior) apply_cell_piped<FMT>(''', '''  {
    const uint32_t x = cur.yrow[0] ^ cur.yrow[1] ^ cur.yrow[2] ^ cur.yrow[3] ^ cur.uu[0] ^ cur.uu[1] ^ cur.vv[0] ^ cur.vv[1] ^ __float_as_uint(e1 + e2 + e3 + e4);
    for (int oy = 0; oy < 4; ++oy) st_stream(reinterpret_cast<uint4*>(static_cast<char*>(im.dst) + ((4u * cy + oy) * c.width + 4u * cx) * 4u), make_uint4(x, x + oy, x ^ 1u, x ^ 2u));
    cx = ncx; cy = ncy;
    return more;
  }
  if (T::kOetf) {
    if (interior) apply_cell_piped<FMT>(''')
    # the prologue: compiled out for the XN builds (the tables are never read by the floor's cell)
    s = sub(s, '''  {
    // all loads first, then all stores: one trip through L2's latency per block instead of one per piece
    constexpr uint32_t kN1 = T::kS1Bytes / 16u''', '''  if (!UHDR_NOPRO) {
    // all loads first, then all stores: one trip through L2's latency per block instead of one per piece
    constexpr uint32_t kN1 = T::kS1Bytes / 16u''')
    s = sub(s, '''  __shared__ uint4 s_tab[T::kBytes / 16u];''', '''  __shared__ uint4 s_tab[UHDR_NOPRO ? 1u : T::kBytes / 16u];''')
    open(p, "w").write(s)
    procs = []
    for name in want:
        nopro, cpt = (1 if name[1] == "N" else 0), int(name[2:])
        out = os.path.join(ROOT, "scripts", "ab", "libvar_%s.so" % name)
        procs.append((name, subprocess.Popen(["/opt/rocm/bin/hipcc"] + FLAGS + ["-DUHDR_NOPRO=%d" % nopro, "-DUHDR_CPT=%d" % cpt, "-shared", "-o", out] + SRCS,
                                             cwd=TMP + "/csrc", stderr=subprocess.PIPE)))
    for name, pr in procs:
        err = pr.communicate()[1].decode()
        assert pr.returncode == 0, (name, [l for l in err.splitlines() if "error" in l][:5])
    print("built", " ".join("scripts/ab/libvar_%s.so" % n for n in want))


if __name__ == "__main__":
    main()
