P='import json,sys; d=json.loads(sys.stdin.read()); print({k:(v["generate_us"], v["apply_hlg_us"], v["apply_pq_us"]) for k,v in d.items()})'
for i in 1 2 3; do
echo A; UHDR_HIP_LIB=$PWD/scripts/ab/libvar_A.so python scripts/time_single.py 2>/dev/null | python -c "$P"
echo B; UHDR_HIP_LIB=$PWD/scripts/ab/libvar_B.so python scripts/time_single.py 2>/dev/null | python -c "$P"
done
VARIANTS="A B" STEPS=100 ROUNDS=4 bash scripts/ab/run_ab.sh
