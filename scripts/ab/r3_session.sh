set -x
bash scripts/final_refresh.sh r03 > gpurun_out/final_refresh_r03.log 2>&1
tail -8 gpurun_out/final_refresh_r03.log
cat gpurun_out/pmc_apply_progress_r03.log
python scripts/clock_trace.py 3 > gpurun_out/r03_clock_trace.txt 2>&1
bash scripts/kernel_times_of.sh scripts/time_single.py > gpurun_out/r03_single_kernel_times.txt 2>&1
grep "k_apply\|k_generate" gpurun_out/r03_single_kernel_times.txt
