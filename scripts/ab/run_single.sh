for r in 1 2; do for v in ${VARIANTS:-A B}; do
UHDR_HIP_LIB=$PWD/scripts/ab/libvar_$v.so python scripts/time_single.py | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', {k:(v['generate_us'], v['apply_hlg_us']) for k,v in d.items()})"
done; done
