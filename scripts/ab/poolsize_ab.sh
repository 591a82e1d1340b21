#!/bin/bash
# bench.py with placement pools of several sizes, interleaved: first on the box as it comes, then after other processes have used its memory
one() { python bench.py --steps 30 --warmup 5 --pool-gib $1 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2 pool $1 GiB: value', d['value'], 'fixed', d['fixed_batch']['value'], 'hipmalloc', d['placement']['hipmalloc']['value'], 'generate', d['kernels']['generate']['avg_launch_ms'], 'apply', d['kernels']['apply']['avg_launch_ms'], 'cold', d['cold_start']['value'])"; }
for round in 1 2; do for g in ${SIZES:-0 -1}; do one $g fresh; done; done
python -m pytest tests/test_gpu_parity.py tests/test_gpu_lut.py -x -q -m gpu > /dev/null 2>&1
for round in 1 2; do for g in ${SIZES:-0 -1}; do one $g churned; done; done
