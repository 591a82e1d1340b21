"""Build the gfx950 shared library (libuhdr_hip.so) in-tree with hipcc.

    python -m libultrahdr_dev_amd.build            # build if stale
    python -m libultrahdr_dev_amd.build --force

hipcc cross-compiles for gfx950 without a GPU present.  -ffp-contract=off is a parity requirement
(the reference's x86-64 build has no FMA; see csrc/uhdr_device_math.h).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libuhdr_hip.so")
SHIM_LIB = os.path.join(HERE, "libultrahdr_shim.so")
COMM_LIB = os.path.join(HERE, "libuhdr_hip_comm.so")   # include/uhdr_hip_comm.h: the RCCL side, apart from the pixel path

SOURCES = ["uhdr_kernels.hip", "uhdr_capi.hip", "uhdr_jpeg.hip", "uhdr_jpeg_dec.hip", "uhdr_jpeg_hdr.cpp", "uhdr_jpeg_prog.cpp", "uhdr_jpegr.cpp"]
DEPS = SOURCES + ["uhdr_kernels.h", "uhdr_device_math.h", "uhdr_jpeg.h", "uhdr_jpegr.h", os.path.join(ROOT, "include", "uhdr_hip.h")]
COMM_SOURCES = ["uhdr_comm.hip"]
COMM_DEPS = COMM_SOURCES + [os.path.join(ROOT, "include", "uhdr_hip_comm.h")]
SHIM_SOURCES = ["ultrahdr_shim.cpp"]
SHIM_DEPS = SHIM_SOURCES + [os.path.join(ROOT, "include", "uhdr_hip.h"),
                            os.path.join(ROOT, "include", "ultrahdr_hip", "ultrahdr_hip.h")]

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def _stale(target, deps, base):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    for d in deps:
        p = d if os.path.isabs(d) else os.path.join(base, d)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build(force=False, verbose=False):
    """Compile csrc/*.hip -> libuhdr_hip.so, csrc/uhdr_comm.hip -> libuhdr_hip_comm.so and the C++ shim -> libultrahdr_shim.so."""
    if force or _stale(LIB, DEPS, CSRC):
        cmd = [HIPCC] + FLAGS + ["-shared", "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    shim_src = [os.path.join(CSRC, s) for s in SHIM_SOURCES]
    if all(os.path.exists(s) for s in shim_src) and (force or _stale(SHIM_LIB, SHIM_DEPS, CSRC)):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
               "-o", SHIM_LIB] + shim_src + ["-L" + HERE, "-luhdr_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    # last, and not fatal: the pixel path does not depend on RCCL, so neither does its build.  Without librccl the comm library
    # is skipped (load_comm() then raises its ImportError; nothing else needs it)
    if force or _stale(COMM_LIB, COMM_DEPS, CSRC):
        rocm_lib = os.path.join(os.path.dirname(os.path.dirname(HIPCC)), "lib")
        cmd = [HIPCC] + FLAGS + ["-shared", "-o", COMM_LIB] + [os.path.join(CSRC, s) for s in COMM_SOURCES] + ["-L" + rocm_lib, "-lrccl", "-Wl,-rpath," + rocm_lib]
        if verbose:
            print(" ".join(cmd))
        try:
            subprocess.check_call(cmd)
        except (subprocess.CalledProcessError, OSError) as e:
            print("libuhdr_hip_comm.so skipped (is RCCL installed under %s?): %s" % (rocm_lib, e), file=sys.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
