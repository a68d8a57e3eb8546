#!/bin/bash
# kernel stats of one Lloyd run per library
for v in shipped bnd512 bnd1024; do
  if [ $v = shipped ]; then export BROV2_LIBRARY=$PWD/bluerov2_dynamics_amd/libbrov2.so; else export BROV2_LIBRARY=$PWD/build_variants/$v/libbrov2.so; fi
  echo "== $v"
  tools/kstats_run.sh gpurun_out/ks_$v -- python3 $PWD/tools/time_lloyd_ab.py 300 0 | grep -E "bounds|assign_lds_kernel<12, true|cdist|mstep" || exit 1
  grep variant gpurun_out/ks_$v/trace.log
done
