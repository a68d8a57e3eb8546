#!/bin/bash
# GPU box: Lloyd (300 iterations, 1e7 x 12, k = 512) with the experiments library's run-time knobs: tools/sweep_lloyd_knobs.sh NAME v1 v2 ...
#   e.g. tools/sweep_lloyd_knobs.sh BROV2_KM_SORT_MOVED 0.04 0.06 0.08
export BROV2_LIBRARY=$PWD/build_variants/experiments/libbrov2.so
name=$1; shift
for v in "$@"; do
  echo "== $name=$v"
  env $name=$v timeout -k 10 200 python3 tools/time_lloyd_ab.py 300 0 2>&1 | grep variant
done
