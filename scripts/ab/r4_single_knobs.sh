#!/bin/bash
# the single-image FAST apply calls under other launch rules (scripts/ab/apply_launch_knobs.py), one box
export UHDR_HIP_LIB=$PWD/scripts/ab/libvar_K.so
for rep in 1 2; do
for knobs in "A=shipped" "UHDR_X_MINBLK=900" "UHDR_X_MINBLK=1800" "UHDR_X_MINBLK=3600" "UHDR_X_MINBLK=100000" "UHDR_X_MINBLK=300" "UHDR_X_MINBLK=900 UHDR_X_CPT=4" "UHDR_X_MINBLK=224"; do
  echo -n "$knobs  "; env $knobs python scripts/time_single_formats.py 2>/dev/null
done; done
echo "--no-stats:"; unset UHDR_HIP_LIB
python bench.py --steps 60 --no-cpu-baseline --no-other-configs --no-stats 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no-stats', d['value'], d['kernels']['generate']['avg_launch_ms'], d['kernels']['apply']['avg_launch_ms'])"
python bench.py --steps 60 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stats   ', d['value'], d['kernels']['generate']['avg_launch_ms'], d['kernels']['apply']['avg_launch_ms'])"
