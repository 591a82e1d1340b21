#!/bin/bash
# what the HIP event records around the launches cost the timed steps (one box, interleaved): value / ms per step / fixed-batch value
for rep in 1 2 3; do for e in all apply none; do
python bench.py --events $e --steps 40 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('events=$e', d['value'], d['ms_per_step'], d['fixed_batch']['value'], d['fixed_batch']['ms_per_step'], d['cold_start']['value'], d['kernels']['apply']['avg_launch_ms'])"
done; done
