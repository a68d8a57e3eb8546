#!/bin/bash
# GPU box: kernel stats of one Lloyd run (tools/time_lloyd_ab.py, 300 iterations at 1e7 x 12, k = 512) per library:
#   tools/ab_lloyd_kernels.sh [variant ...]     the shipped library and build_variants/<variant>/libbrov2.so
for v in shipped "$@"; do
  if [ $v = shipped ]; then export BROV2_LIBRARY=$PWD/bluerov2_dynamics_amd/libbrov2.so; else export BROV2_LIBRARY=$PWD/build_variants/$v/libbrov2.so; fi
  echo "== $v"
  tools/kstats_run.sh gpurun_out/ks_$v -- python3 $PWD/tools/time_lloyd_ab.py 300 0 4 | grep -E "bounds|assign_lds_kernel<12|cdist|mstep" || exit 1
  grep variant gpurun_out/ks_$v/trace.log
done
