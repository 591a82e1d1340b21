#!/usr/bin/env python3
"""Store-organisation floors of k_apply_s4 for same-box A/B runs (never shipped; results are NOT the reference's bytes): the walk's
loads and stores with one XOR in between (variant X1 of apply_floor_variants.py) and the same with the block's 4 x 8 KiB of output
passed through LDS and stored

  X1  by every lane for its own cell (the shipped pattern: a wave stores 4 rows x 1 KiB)
  S0  by wave 0 alone (32 stores of 1 KiB per cell step, in address order): ONE storing wave per block
  S2  by waves 0 and 4 (16 stores each): two storing waves per block
  SC  by every wave, but 4 KiB contiguous per wave (row w / 2, half w % 2) instead of 4 rows x 1 KiB
  SW  X1 + every wave waits for its own stores (s_waitcnt vmcnt(0)) before it issues the next cell's: at most 4 stores per wave in flight

scripts/ab/xcd_affinity.hip showed that pure fills run faster the FEWER waves store at a time (256 blocks x 4 waves: 6.4 TB/s;
2048 x 4: 4.4) -- this asks whether the same holds inside the kernel, beside its loads.   VARIANTS="X1 S0 S2 SC SW" run_ab.sh
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
TMP = "/tmp/uhdr_store/a/b"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function"]
SRCS = ["uhdr_kernels.hip", "uhdr_capi.hip", "uhdr_jpeg.hip", "uhdr_jpeg_dec.hip", "uhdr_jpeg_hdr.cpp", "uhdr_jpeg_prog.cpp", "uhdr_jpegr.cpp"]
MODES = {"X1": 1, "S0": 4, "S2": 6, "SC": 5, "SW": 7}


def sub(s, old, new):
    assert old in s, old[:80]
    return s.replace(old, new, 1)


def main():
    want = sys.argv[1:] or list(MODES)
    shutil.rmtree("/tmp/uhdr_store", ignore_errors=True)
    os.makedirs(TMP)
    shutil.copytree(os.path.join(ROOT, "libultrahdr_dev_amd", "csrc"), TMP + "/csrc")
    shutil.copytree(os.path.join(ROOT, "include"), "/tmp/uhdr_store/a/include")
    p = TMP + "/csrc/uhdr_kernels.hip"
    s = open(p).read()
    s = sub(s, '#include "uhdr_kernels.h"\n', '#include "uhdr_kernels.h"\n#ifndef UHDR_XP\n#define UHDR_XP 0\n#endif\n')
    s = sub(s, '''  __shared__ uint4 s_xch[FMT == 1 ? (kApplyBlock / 64) * kXchPerWave : 1];''',
            '''  __shared__ uint4 s_xch[UHDR_XP >= 4 && UHDR_XP != 7 ? 4 * kApplyBlock : FMT == 1 ? (kApplyBlock / 64) * kXchPerWave : 1];''')
    s = sub(s, '''  if (T::kOetf) {
    if (interior) apply_cell_piped<FMT>(''', '''#if UHDR_XP != 0
  {
    const uint32_t x = cur.yrow[0] ^ cur.yrow[1] ^ cur.yrow[2] ^ cur.yrow[3] ^ cur.uu[0] ^ cur.uu[1] ^ cur.vv[0] ^ cur.vv[1] ^ __float_as_uint(e1 + e2 + e3 + e4);
    char* dst = static_cast<char*>(im.dst);
#if UHDR_XP == 1 || UHDR_XP == 7
    for (int oy = 0; oy < 4; ++oy) st_stream(reinterpret_cast<uint4*>(dst + ((4u * cy + oy) * c.width + 4u * cx) * 4u), make_uint4(x, x + oy, x ^ 1u, x ^ 2u));
#if UHDR_XP == 7
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (memory operations retire in order: this also waits for the next cell's loads, which the next cell needs anyway)
#endif
#else
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (int oy = 0; oy < 4; ++oy) s_xch[oy * kApplyBlock + tid] = make_uint4(x, x + oy, x ^ 1u, x ^ 2u);
    __syncthreads();
    const uint32_t idx0 = cy * c.map_w + cx - tid;   // the chunk's first cell
#if UHDR_XP == 4
    if (wave == 0u)
      for (uint32_t j = 0; j < 32u; ++j) {
        const uint32_t row = j >> 3, t2 = (j & 7u) * 64u + lane, i2 = idx0 + t2, y2 = i2 / c.map_w, x2 = i2 - y2 * c.map_w;
        const uint4 v = s_xch[row * kApplyBlock + t2];
        if (y2 < c.map_h) st_stream(reinterpret_cast<uint4*>(dst + ((4u * y2 + row) * c.width + 4u * x2) * 4u), v);
      }
#elif UHDR_XP == 6
    if ((wave & 3u) == 0u)
      for (uint32_t j = 0; j < 16u; ++j) {
        const uint32_t jj = j + (wave >> 2) * 16u, row = jj >> 3, t2 = (jj & 7u) * 64u + lane, i2 = idx0 + t2, y2 = i2 / c.map_w, x2 = i2 - y2 * c.map_w;
        const uint4 v = s_xch[row * kApplyBlock + t2];
        if (y2 < c.map_h) st_stream(reinterpret_cast<uint4*>(dst + ((4u * y2 + row) * c.width + 4u * x2) * 4u), v);
      }
#else
    for (uint32_t j = 0; j < 4u; ++j) {
      const uint32_t row = wave >> 1, t2 = (wave & 1u) * 256u + j * 64u + lane, i2 = idx0 + t2, y2 = i2 / c.map_w, x2 = i2 - y2 * c.map_w;
      const uint4 v = s_xch[row * kApplyBlock + t2];
      if (y2 < c.map_h) st_stream(reinterpret_cast<uint4*>(dst + ((4u * y2 + row) * c.width + 4u * x2) * 4u), v);
    }
#endif
    __syncthreads();
#endif
    cx = ncx; cy = ncy;
    return more;
  }
#endif
  if (T::kOetf) {
    if (interior) apply_cell_piped<FMT>(''')
    open(p, "w").write(s)
    procs = []
    for name in want:
        out = os.path.join(ROOT, "scripts", "ab", "libvar_%s.so" % name)
        procs.append((name, subprocess.Popen(["/opt/rocm/bin/hipcc"] + FLAGS + ["-DUHDR_XP=%d" % MODES[name], "-shared", "-o", out] + SRCS, cwd=TMP + "/csrc",
                                             stderr=subprocess.PIPE)))
    for name, pr in procs:
        err = pr.communicate()[1].decode()
        assert pr.returncode == 0, (name, [l for l in err.splitlines() if "error" in l][:5])
    print("built", " ".join("scripts/ab/libvar_%s.so" % n for n in want))


if __name__ == "__main__":
    main()
