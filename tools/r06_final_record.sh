#!/bin/bash
# GPU box, repo root: tools/r06_final_record.sh  -- the part of round 6's record that is not a rocprofv3 run of bench.py (tools/profile_round.sh r06,
# tools/r04_lloyd_pmc.sh r06_lloyd, tools/r04_fit_pmc.sh r06_fit are separate calls): counters of config 4's two kernels, where the first
# fit() of a process goes in the three runtime modes, the linear multistep path, the GPU test suite on the shipped and on the experiments
# library, the randomised parity sweep on both, and the 2-rank rehearsal of the bench through the compact line.
set -o pipefail
mkdir -p gpurun_out/r06_final
o=gpurun_out/r06_final
part=${1:-all}
if [ "$part" = "a" ] || [ "$part" = "all" ]; then
timeout -k 10 500 tools/r06_cfg4_pmc.sh r06 > $o/cfg4_pmc.log 2>&1 || echo "cfg4 pmc failed"
{ for m in auto 0 1; do BROV2_TORCH=$m python3 tools/time_first_fit.py 2>&1 | grep -v amdgpu.ids; echo; done; } > $o/fit_time.txt
python3 tools/time_multistep_linear.py 2>&1 | grep -v amdgpu.ids > $o/multistep_linear.txt
python3 tools/time_upload.py 2>&1 | grep -v amdgpu.ids > $o/upload.txt
echo "timings done"
timeout -k 10 900 python -m pytest tests -m gpu -q > $o/gpu_tests.log 2>&1; tail -2 $o/gpu_tests.log
python bench.py --steps 20 --warmup 5 --details $o/bench_details.json > $o/bench.json 2> $o/bench.err; echo "bench rc=$? bytes=$(wc -c < $o/bench.json)"
fi
if [ "$part" = "b" ] || [ "$part" = "all" ]; then
BROV2_LIBRARY=$PWD/build_variants/experiments/libbrov2.so timeout -k 10 900 python -m pytest tests -m gpu -q > $o/gpu_tests_experiments.log 2>&1; tail -2 $o/gpu_tests_experiments.log
timeout -k 10 400 python3 tests/stress_parity.py 30 6 > $o/stress_parity.log 2>&1; tail -2 $o/stress_parity.log
BROV2_LIBRARY=$PWD/build_variants/experiments/libbrov2.so timeout -k 10 400 python3 tests/stress_parity.py 30 7 > $o/stress_experiments.log 2>&1; tail -2 $o/stress_experiments.log
BROV2_BENCH_SHARE_GPU=1 BROV2_BENCH_BACKEND=gloo timeout -k 10 900 python bench.py --gpus 2 --steps 20 --warmup 5 --details $o/rehearsal_2ranks_details.json > $o/rehearsal_2ranks.json 2> $o/rehearsal_2ranks.err; echo "rehearsal rc=$? bytes=$(wc -c < $o/rehearsal_2ranks.json) lines=$(wc -l < $o/rehearsal_2ranks.json)"
fi
