#!/bin/bash
# scripts/isa_of.sh MANGLED_SUBSTRING [-DFLAG...]: gfx950 assembly of one kernel of csrc/uhdr_kernels.hip -> /tmp/isa_<substring>.s
K=$1; shift
cd "$(dirname "$0")/../libultrahdr_dev_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math "$@" --cuda-device-only -S uhdr_kernels.hip -o /tmp/isa_all.s 2>/dev/null || exit 1
awk -v k="$K" 'index($0, k) && /^_Z[^ ]*:/ {f=1} f{print} f && /\.amdhsa_kernel/ {exit}' /tmp/isa_all.s > "/tmp/isa_$K.s"
wc -l "/tmp/isa_$K.s"
