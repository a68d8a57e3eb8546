#!/bin/bash
# GPU box, repo root: tools/r05_profile.sh  -- the round's measurement record under gpurun_out/prof_r05*/ (copy what is judged into profiles/):
#   prof_r05/        default bench.py, rocprofv3 --kernel-trace --stats of the same command, kernel times without first launches, one --pmc
#                    pass per counter group (kernel trace only), pmc_summary.json, the FETCH_SIZE / WRITE_SIZE known-bytes probe
#   r05_lloyd/       SQ / HBM counters of the E-step kernel over the shipped Lloyd loop (tools/time_lloyd.py)
#   r05_fit/         kernel stats + counters of the fit() path without centres (tools/time_fit.py)
set -e -o pipefail
tools/profile_round.sh r05
echo "== lloyd"
tools/r04_lloyd_pmc.sh r05_lloyd
echo "== fit"
tools/r04_fit_pmc.sh r05_fit
