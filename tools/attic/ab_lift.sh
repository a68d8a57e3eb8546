#!/bin/bash
for lib in bluerov2_dynamics_amd/libbrov2.so build_variants/*/libbrov2.so; do
  echo "== $lib"; BROV2_LIBRARY=$PWD/$lib python tools/time_lift.py 2>&1 | grep "lift+gram" | tail -2
done
