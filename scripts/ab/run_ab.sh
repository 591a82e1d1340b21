#!/bin/bash
# same-box A/B of library variants (scripts/ab/libvar_<name>.so): each variant benched ROUNDS times, interleaved; prints
# value, generate ms, apply ms per run and the medians at the end.   VARIANTS="A B" STEPS=100 ROUNDS=4 bash scripts/ab/run_ab.sh
V=${VARIANTS:-A B}
for round in $(seq 1 ${ROUNDS:-3}); do
for v in $V; do
  UHDR_HIP_LIB=$PWD/scripts/ab/libvar_$v.so python bench.py --steps ${STEPS:-50} --warmup 5 --no-cpu-baseline --no-other-configs $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['kernels']['generate']['avg_launch_ms'], d['kernels']['apply']['avg_launch_ms'], d['cold_start']['value'])"
done
done | tee /tmp/ab_runs.txt
python - <<'PY'
import statistics as st
rows = [l.split() for l in open('/tmp/ab_runs.txt') if l.strip()]
for v in sorted({r[0] for r in rows}):
    r = [x for x in rows if x[0] == v]
    print('median', v, 'value %.0f' % st.median(float(x[1]) for x in r), 'generate %.4f' % st.median(float(x[2]) for x in r), 'apply %.4f' % st.median(float(x[3]) for x in r))
PY
if [ -n "$SINGLE" ]; then for r in 1 2; do for v in $V; do
UHDR_HIP_LIB=$PWD/scripts/ab/libvar_$v.so python scripts/time_single.py | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', {k:(v['generate_us'], v['apply_hlg_us'], v['apply_pq_us']) for k,v in d.items()})"
done; done; fi
