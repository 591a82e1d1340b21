#!/bin/bash
# same-box A/B of library variants: each variant benched twice, interleaved
for round in 1 2; do
for v in ${VARIANTS:-A B}; do
  UHDR_HIP_LIB=$PWD/scripts/ab/libvar_$v.so python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-other-configs $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['kernels']['generate']['avg_launch_ms'], d['kernels']['apply']['avg_launch_ms'])"
done
done
