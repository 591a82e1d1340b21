#!/bin/bash
# scripts/ab/build_var.sh NAME [-DFLAG=...]: a variant of the library for same-box A/B runs (scripts/ab/run_ab.sh)
N=$1; shift
cd "$(dirname "$0")/../../libultrahdr_dev_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function "$@" -shared -o ../../scripts/ab/libvar_$N.so uhdr_kernels.hip uhdr_capi.hip uhdr_jpeg.hip uhdr_jpeg_dec.hip uhdr_jpeg_hdr.cpp uhdr_jpeg_prog.cpp uhdr_jpegr.cpp 2>&1 | grep -v "warning\|^ *[0-9]* |\|\^\|generated" | head
