#!/usr/bin/env python3
"""Cache-policy builds of the two streaming kernels for same-box A/B runs (never shipped): a copy of csrc/ under /tmp with

  GT   generate loads the 8-bit Y / U / V planes with PLAIN loads (the P010 planes stay non-temporal): the planes apply reads
       again may then still be in the 256 MB Infinity Cache when apply of the same chunk of frames runs
  GTA  GT + apply loads its Y rows with plain loads as well (they are non-temporal in the shipped kernel)
  AN   apply loads Y, U, V all non-temporally (the last use of those bytes inside a step)
  GTAN GT + AN: first use temporal, last use non-temporal

  R / GTR / GTANR   the shipped kernels / GT / GTAN with apply walking every image from its last span of cells to its first (what generate
       read last is what apply then reads first: the tail of the 8-bit planes may still be on-die WITHIN a step)

compiled into scripts/ab/libvar_<name>.so.  VERDICT r03 item 1(b); scripts/time_step_pipeline.py runs them, profiles/r04_pipeline_ab.txt
has the numbers.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
TMP = "/tmp/uhdr_policy/a/b"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function"]
SRCS = ["uhdr_kernels.hip", "uhdr_capi.hip", "uhdr_jpeg.hip", "uhdr_jpeg_dec.hip", "uhdr_jpeg_hdr.cpp", "uhdr_jpeg_prog.cpp", "uhdr_jpegr.cpp"]


def sub(s, old, new, count=1):
    assert s.count(old) >= count, old[:80]
    return s.replace(old, new, count)


GEN_T = [
    ("const uint2 p = ld_stream(reinterpret_cast<const uint2*>(im.y + (yoff + r * im.y_stride)));",
     "const uint2 p = *reinterpret_cast<const uint2*>(im.y + (yoff + r * im.y_stride));"),
    ("const uint32_t uu = ld_stream(reinterpret_cast<const uint32_t*>(im.u + (coff + r * im.c_stride)));",
     "const uint32_t uu = *reinterpret_cast<const uint32_t*>(im.u + (coff + r * im.c_stride));"),
    ("const uint32_t vv = ld_stream(reinterpret_cast<const uint32_t*>(im_v + (coff + r * im.c_stride)));",
     "const uint32_t vv = *reinterpret_cast<const uint32_t*>(im_v + (coff + r * im.c_stride));"),
]
# (apply_load_cell_pk: the walk's loads)
APP_Y_PLAIN = [
    ("""struct __attribute__((packed)) U16Any { uint16_t v; };
// (mx < map_w - 1: the column of the byte pairs; cy1: the row of the lower taps)
__device__ __forceinline__ void apply_load_cell_pk(const AppConsts& c, const AppImage& im, uint32_t cx, uint32_t cy, uint32_t mx, uint32_t cy1, ApplyCellPk& o) {
  const uint32_t yoff = 4u * cy * im.y_stride + 4u * cx;
#pragma unroll
  for (int r = 0; r < 4; ++r) o.yrow[r] = ld_stream(reinterpret_cast<const uint32_t*>(im.y + (yoff + r * im.y_stride)));""",
     """struct __attribute__((packed)) U16Any { uint16_t v; };
// (mx < map_w - 1: the column of the byte pairs; cy1: the row of the lower taps)
__device__ __forceinline__ void apply_load_cell_pk(const AppConsts& c, const AppImage& im, uint32_t cx, uint32_t cy, uint32_t mx, uint32_t cy1, ApplyCellPk& o) {
  const uint32_t yoff = 4u * cy * im.y_stride + 4u * cx;
#pragma unroll
  for (int r = 0; r < 4; ++r) o.yrow[r] = *reinterpret_cast<const uint32_t*>(im.y + (yoff + r * im.y_stride));"""),
]
APP_C_NT = [
    ("""    o.uu[r] = *reinterpret_cast<const uint16_t*>(im.u + (coff + r * im.c_stride));
    o.vv[r] = *reinterpret_cast<const uint16_t*>(im.v + (coff + r * im.c_stride));
  }
  o.mrow[0] = reinterpret_cast<const U16Any*>(im.map + (cy * c.map_w + mx))->v;""",
     """    o.uu[r] = __builtin_nontemporal_load(reinterpret_cast<const uint16_t*>(im.u + (coff + r * im.c_stride)));
    o.vv[r] = __builtin_nontemporal_load(reinterpret_cast<const uint16_t*>(im.v + (coff + r * im.c_stride)));
  }
  o.mrow[0] = reinterpret_cast<const U16Any*>(im.map + (cy * c.map_w + mx))->v;"""),
]
# apply walks every image from its LAST span of cells to its first: what generate (whose blocks sweep the images top to bottom,
# all images together) read last -- and, with GT, left in the Infinity Cache -- is what apply then reads first
APP_REVERSE = [
    ("""  const uint32_t img_i = blockIdx.x, span = blockIdx.y;
  const AppImage& im = b.img[img_i];
  const uint32_t idx = span * c.cells_per_thread * kApplyBlock + threadIdx.x;""",
     """  const uint32_t img_i = blockIdx.x, span = gridDim.y - 1u - blockIdx.y;
  const AppImage& im = b.img[img_i];
  const uint32_t idx = span * c.cells_per_thread * kApplyBlock + threadIdx.x;"""),
]
VARIANTS = {"GT": GEN_T, "GTA": GEN_T + APP_Y_PLAIN, "AN": APP_C_NT, "GTAN": GEN_T + APP_C_NT,
            "R": APP_REVERSE, "GTR": GEN_T + APP_REVERSE, "GTANR": GEN_T + APP_C_NT + APP_REVERSE}


def main():
    want = sys.argv[1:] or list(VARIANTS)
    shutil.rmtree("/tmp/uhdr_policy", ignore_errors=True)
    procs = []
    for name in want:
        d = "/tmp/uhdr_policy/%s/b" % name
        os.makedirs(d)
        shutil.copytree(os.path.join(ROOT, "libultrahdr_dev_amd", "csrc"), d + "/csrc")
        shutil.copytree(os.path.join(ROOT, "include"), "/tmp/uhdr_policy/%s/include" % name)
        p = d + "/csrc/uhdr_kernels.hip"
        s = open(p).read()
        for old, new in VARIANTS[name]:
            s = sub(s, old, new)
        open(p, "w").write(s)
        out = os.path.join(ROOT, "scripts", "ab", "libvar_%s.so" % name)
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc"] + FLAGS + ["-shared", "-o", out] + SRCS, cwd=d + "/csrc", stderr=subprocess.DEVNULL))
    for pr in procs:
        assert pr.wait() == 0
    print("built", " ".join("scripts/ab/libvar_%s.so" % n for n in want))


if __name__ == "__main__":
    main()
