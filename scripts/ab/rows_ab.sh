#!/bin/bash
# same-box A/B of FAST apply's two forms inside ONE library (UHDR_HIP_APPLY_ROWS: 0 the walk, 1 one-shot rows image by image,
# 2 rows with the images of the launch interleaved): value, generate ms, apply ms per run, medians at the end.
for round in $(seq 1 ${ROUNDS:-3}); do
for v in ${POLICIES:-0 1 2}; do
  UHDR_HIP_APPLY_ROWS=$v python bench.py --steps ${STEPS:-60} --warmup 5 --no-cpu-baseline --no-other-configs $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rows=$v', d['value'], d['kernels']['generate']['avg_launch_ms'], d['kernels']['apply']['avg_launch_ms'], d['fixed_batch']['value'], d['cold_start']['value'])"
done
done | tee /tmp/rows_ab.txt
python - <<'PY'
import statistics as st
rows = [l.split() for l in open('/tmp/rows_ab.txt') if l.strip()]
for v in sorted({r[0] for r in rows}):
    r = [x for x in rows if x[0] == v]
    print('median', v, 'value %.0f' % st.median(float(x[1]) for x in r), 'generate %.4f' % st.median(float(x[2]) for x in r), 'apply %.4f' % st.median(float(x[3]) for x in r))
PY
