#!/bin/bash
# ONE box: the profile passes (kernel trace with the traced process's own line, PMC passes -> HBM traffic, stamped with the kernel
# sources), then the plain bench line (which reports that traffic), the apply counters, and every fraction recomputed from those
# files -> gpurun_out/*_$1.*   (scripts/final_refresh.sh TAG; copy to profiles/)
T=$1
bash scripts/profile_bench.sh $T > gpurun_out/prof_$T.log 2>&1
cp gpurun_out/prof_${T}_traffic.json profiles/traffic_latest.json
python bench.py > gpurun_out/bench_$T.json 2> gpurun_out/bench_$T.err
bash scripts/profile_apply_pmc.sh $T > /dev/null 2>&1
python scripts/recompute_fractions.py gpurun_out/bench_$T.json gpurun_out/prof_$T.json > gpurun_out/onebox_$T.txt 2>&1
cat gpurun_out/onebox_$T.txt
