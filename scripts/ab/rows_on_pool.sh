#!/bin/bash
# the one-shot rows form of FAST apply (scripts/ab/patches/r04_apply_one_shot_rows.patch, built as libvar_ROWS.so) on pool memory
one() { UHDR_HIP_LIB=$PWD/scripts/ab/libvar_ROWS.so env $2 python bench.py --steps 40 --warmup 5 --no-placement-ab --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'fixed', d['fixed_batch']['value'], 'generate', d['kernels']['generate']['avg_launch_ms'], 'apply', d['kernels']['apply']['avg_launch_ms'])"; }
for round in 1 2 3; do
  one walk UHDR_HIP_APPLY_ROWS=0
  one rows-image-by-image UHDR_HIP_APPLY_ROWS=1
  one rows-images-interleaved UHDR_HIP_APPLY_ROWS=2
done
