#!/bin/bash
# same-box A/B of LUT-mode apply (scripts/time_lut.py) between library variants, interleaved:  VARIANTS="OLD NEW" bash scripts/ab/lut_ab.sh
for round in 1 2 3; do for v in ${VARIANTS:-OLD NEW}; do
  UHDR_HIP_LIB=$PWD/scripts/ab/libvar_$v.so python scripts/time_lut.py 2>/dev/null | grep "32 frame" | sed "s/^/$v  /"
done; done
