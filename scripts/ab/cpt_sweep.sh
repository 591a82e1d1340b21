P='import json,sys; d=json.loads(sys.stdin.read()); print({k:(v["generate_us"], v["apply_hlg_us"], v["apply_pq_us"]) for k,v in d.items()})'
run() { echo "$@"; env "$@" UHDR_HIP_LIB=$PWD/scripts/ab/libvar_X.so python scripts/time_single.py 2>/dev/null | python -c "$P"; }
for i in 1 2; do
run A=default
run UHDR_X_EDGE=0
run UHDR_X_EDGE=1
run UHDR_X_EDGE=1 UHDR_X_ER=2
run UHDR_X_EDGE=1 UHDR_X_CPT=4
run UHDR_X_EDGE=1 UHDR_X_CPT=16
run UHDR_X_EDGE=0 UHDR_X_CPT=4
done
