#!/bin/bash
# round 4, VERDICT r03 item 1: rotating-batch protocol, then the chunked two-stream pipeline under each cache policy (one box)
mkdir -p gpurun_out
{
for v in B GT GTAN; do
  UHDR_HIP_LIB=$PWD/scripts/ab/libvar_$v.so python scripts/time_step_pipeline.py 3 || exit 1
done
for v in GTA AN B; do
  CONFIGS=64x1,16x1,8x2 UHDR_HIP_LIB=$PWD/scripts/ab/libvar_$v.so python scripts/time_step_pipeline.py 3 || exit 1
done
} 2>&1 | grep -v Warning > gpurun_out/r4_pipeline.txt
python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/r4_bench_rot.json 2> gpurun_out/r4_bench_rot.err
