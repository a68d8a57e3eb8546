#!/bin/bash
# GPU box: time 300 Lloyd iterations with every library under build_variants/ (and the shipped one) -> gpurun_out/lloyd_variants.txt
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/lloyd_variants.txt
: > $out
for lib in bluerov2_dynamics_amd/libbrov2.so build_variants/*/libbrov2.so; do
  [ -f "$lib" ] || continue
  name=$(basename $(dirname $lib))
  echo "== $name" >> $out
  BROV2_LIBRARY=$PWD/$lib timeout -k 10 200 python3 -u tools/time_lloyd.py 10000000 ${1:-300} 2>&1 | grep -v amdgpu.ids >> $out || exit 1
  tail -2 $out
done
