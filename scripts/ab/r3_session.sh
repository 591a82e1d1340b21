set -x
bash scripts/profile_apply_pmc.sh r03 > /dev/null 2>&1
cat gpurun_out/pmc_apply_progress_r03.log
VARIANTS="A B X1 X2" STEPS=100 ROUNDS=3 bash scripts/ab/run_ab.sh > gpurun_out/r03_apply_floors.log 2>&1
tail -5 gpurun_out/r03_apply_floors.log
python scripts/clock_trace.py 4 > gpurun_out/r03_clock_trace.txt 2>&1
head -5 gpurun_out/r03_clock_trace.txt
