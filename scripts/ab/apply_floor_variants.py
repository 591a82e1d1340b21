#!/usr/bin/env python3
"""Memory-floor builds of k_apply_s4 for same-box A/B runs (never shipped): a copy of csrc/ under /tmp with one of

  X1  the walk's loads and stores, one XOR in between (no arithmetic)        -- what the memory system gives this access pattern
  X2  the whole kernel without its stores (a store behind an impossible test) -- arithmetic + loads
  X3  the whole kernel, every load from the block's first 64 cells            -- arithmetic + stores (the loads hit in L2 / L1)

compiled into scripts/ab/libvar_X<k>.so.    python scripts/ab/apply_floor_variants.py && VARIANTS="B X1 X2 X3" bash scripts/ab/run_ab.sh
(B: cp libultrahdr_dev_amd/libuhdr_hip.so scripts/ab/libvar_B.so).  profiles/r03_apply_memory_floors_ab.txt has the numbers.
"""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
TMP = "/tmp/uhdr_floor/a/b"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function"]
SRCS = ["uhdr_kernels.hip", "uhdr_capi.hip", "uhdr_jpeg.hip", "uhdr_jpeg_dec.hip", "uhdr_jpeg_hdr.cpp", "uhdr_jpeg_prog.cpp", "uhdr_jpegr.cpp"]


def sub(s, old, new):
    assert old in s, old[:80]
    return s.replace(old, new, 1)


def main():
    shutil.rmtree("/tmp/uhdr_floor", ignore_errors=True)
    os.makedirs(TMP)
    shutil.copytree(os.path.join(ROOT, "libultrahdr_dev_amd", "csrc"), TMP + "/csrc")
    shutil.copytree(os.path.join(ROOT, "include"), "/tmp/uhdr_floor/a/include")
    p = TMP + "/csrc/uhdr_kernels.hip"
    s = open(p).read()
    s = sub(s, '#include "uhdr_kernels.h"\n', '#include "uhdr_kernels.h"\n#ifndef UHDR_XP\n#define UHDR_XP 0\n#endif\n')
    s = sub(s, '''  if (T::kOetf) {
    if (interior) apply_cell_piped<FMT>(''', '''#if UHDR_XP == 1
  {
    const uint32_t x = cur.yrow[0] ^ cur.yrow[1] ^ cur.yrow[2] ^ cur.yrow[3] ^ cur.uu[0] ^ cur.uu[1] ^ cur.vv[0] ^ cur.vv[1] ^ __float_as_uint(e1 + e2 + e3 + e4);
    for (int oy = 0; oy < 4; ++oy) st_stream(reinterpret_cast<uint4*>(static_cast<char*>(im.dst) + ((4u * cy + oy) * c.width + 4u * cx) * 4u), make_uint4(x, x + oy, x ^ 1u, x ^ 2u));
    cx = ncx; cy = ncy;
    return more;
  }
#endif
  if (T::kOetf) {
    if (interior) apply_cell_piped<FMT>(''')
    s = sub(s, '''        st_stream(reinterpret_cast<uint4*>(static_cast<char*>(dst) + off), make_uint4(px[0], px[1], px[2], px[3]));''',
            '''#if UHDR_XP == 2
        if ((px[0] ^ px[1] ^ px[2] ^ px[3]) == 0x12345678u)
#endif
        st_stream(reinterpret_cast<uint4*>(static_cast<char*>(dst) + off), make_uint4(px[0], px[1], px[2], px[3]));''')
    s = sub(s, '''  apply_load_cell_pk(c, im, ncx, ncy, ncx - (ncx + 1u == c.map_w ? 1u : 0u), min(ncy + 1u, c.map_h - 1u), nxt);''', '''#if UHDR_XP == 3
  apply_load_cell_pk(c, im, threadIdx.x & 63u, 0u, threadIdx.x & 63u, 1u, nxt);
#else
  apply_load_cell_pk(c, im, ncx, ncy, ncx - (ncx + 1u == c.map_w ? 1u : 0u), min(ncy + 1u, c.map_h - 1u), nxt);
#endif''')
    open(p, "w").write(s)
    procs = []
    for v in (1, 2, 3):
        out = os.path.join(ROOT, "scripts", "ab", "libvar_X%d.so" % v)
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc"] + FLAGS + ["-DUHDR_XP=%d" % v, "-shared", "-o", out] + SRCS, cwd=TMP + "/csrc",
                                      stderr=subprocess.DEVNULL))
    for pr in procs:
        assert pr.wait() == 0
    print("built scripts/ab/libvar_X1.so X2 X3")


if __name__ == "__main__":
    main()
