#!/bin/bash
# round 4, after the placement pools: do the memory policies / spans per block / cells per thread that lost on hipMalloc memory win on pool memory?
one() { UHDR_HIP_LIB=$PWD/scripts/ab/libvar_$1.so env $3 python bench.py --steps 40 --warmup 5 --no-placement-ab --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', d['value'], 'fixed', d['fixed_batch']['value'], 'generate', d['kernels']['generate']['avg_launch_ms'], 'apply', d['kernels']['apply']['avg_launch_ms'])"; }
for round in 1 2 3; do
  one B shipped
  one GT "generate-plain-8bit-loads"
  one AN "apply-nt-chroma"
  one GTAN "GT+AN"
  one R "apply-reverse-walk"
  one T1 "generate-1-span"
  one T2 "generate-2-spans"
  one T8 "generate-8-spans"
  one K "apply-cpt16" UHDR_X_CPT=16
  one K "apply-cpt64" UHDR_X_CPT=64
done
