#!/bin/bash
# one box: profile passes + bench line + apply counters -> gpurun_out/*_$1.*   (scripts/final_refresh.sh TAG)
T=$1
bash scripts/profile_bench.sh $T > gpurun_out/prof_$T.log 2>&1
cp gpurun_out/prof_${T}_traffic.json profiles/traffic_latest.json
python bench.py > gpurun_out/bench_$T.json 2> gpurun_out/bench_$T.err
bash scripts/profile_apply_pmc.sh $T > /dev/null 2>&1
python -c "
import json; d=json.load(open('gpurun_out/bench_$T.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['kernels']['apply']['avg_launch_ms'], d['kernels']['generate']['avg_launch_ms'])"
grep -A5 "== the last" gpurun_out/prof_$T.txt | cut -c1-175
