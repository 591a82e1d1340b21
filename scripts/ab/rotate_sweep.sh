#!/bin/bash
# value / generate ms / apply ms by the number of resident batches the steps rotate over (one box): what does rotation cost, and is it
# the reuse distance (any R > 1 alike) or the footprint (grows with R)?
for rep in 1 2; do for r in 1 2 3 4 6 8; do
python bench.py --rotate $r --steps 40 --no-fixed-batch --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('R=$r', d['value'], d['ms_per_step'], d['kernels']['generate']['avg_launch_ms'], d['kernels']['apply']['avg_launch_ms'], d['cold_start']['value'])"
done; done
for v in AN GT; do for r in 1 3; do
UHDR_HIP_LIB=$PWD/scripts/ab/libvar_$v.so python bench.py --rotate $r --steps 40 --no-fixed-batch --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v R=$r', d['value'], d['ms_per_step'], d['kernels']['generate']['avg_launch_ms'], d['kernels']['apply']['avg_launch_ms'], d['cold_start']['value'])"
done; done
