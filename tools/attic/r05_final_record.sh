#!/bin/bash
set -o pipefail
tools/r05_profile.sh > gpurun_out/r05_profile.log 2>&1 || { echo "profile failed"; tail -5 gpurun_out/r05_profile.log; exit 1; }
echo "profile done"
# Lloyd kernel times (one run under the kernel trace)
tools/kstats_run.sh gpurun_out/ks_final -- python3 $PWD/tools/time_lloyd_ab.py 300 0 > gpurun_out/ks_final.txt 2>&1 || exit 1
python3 tools/kernel_times.py gpurun_out/ks_final/trace/run_kernel_trace.csv > gpurun_out/lloyd_kernel_times.json
echo "lloyd trace done"
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; tail -2 gpurun_out/gpu_tests.log
