#!/bin/bash
# one box: the library's walk floor and one-shot floors (bench step), then the stand-alone floors of the same patterns
VARIANTS="XP32 XO0 XO13 XO37 B" ROUNDS=3 STEPS=60 bash scripts/ab/run_ab.sh > gpurun_out/r4_oneshot_ab.txt 2>&1
scripts/ab/linear_apply_floor 2>&1 | head -24 > gpurun_out/r4_oneshot_standalone.txt
