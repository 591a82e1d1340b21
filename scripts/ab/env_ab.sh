#!/bin/bash
# same-box A/B of runtime environment settings (not library variants): each setting benched ROUNDS times, interleaved.
#   SETTINGS="HIP_FORCE_DEV_KERNARG=0 HIP_FORCE_DEV_KERNARG=1" bash scripts/ab/env_ab.sh
for round in $(seq 1 ${ROUNDS:-4}); do
for v in ${SETTINGS:-HIP_FORCE_DEV_KERNARG=0 HIP_FORCE_DEV_KERNARG=1}; do
  env $v python bench.py --steps ${STEPS:-100} --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['kernels']['generate']['avg_launch_ms'], d['kernels']['apply']['avg_launch_ms'], d['fixed_batch']['value'], d['cold_start']['value'])"
done
done | tee /tmp/env_ab_runs.txt
python - <<'PY'
import statistics as st
rows = [l.split() for l in open('/tmp/env_ab_runs.txt') if l.strip()]
for v in sorted({r[0] for r in rows}):
    r = [x for x in rows if x[0] == v]
    print('median', v, 'value %.0f' % st.median(float(x[1]) for x in r), 'generate %.4f' % st.median(float(x[2]) for x in r), 'apply %.4f' % st.median(float(x[3]) for x in r), 'fixed %.0f' % st.median(float(x[4]) for x in r))
PY
