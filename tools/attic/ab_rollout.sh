#!/bin/bash
# A/B timing of experimental library builds (build_variants/libbrov2_*.so) on the rollout leg of bench.py.
for lib in build_variants/libbrov2_*.so; do
  for rep in 1 2; do
    BROV2_WS=${BROV2_WS:-0} BROV2_LIBRARY=$PWD/$lib timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu --no-edmdc 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$lib', '%.3e' % d['value'], '%.3f ms' % d['roofline']['kernel_ms'])" || exit 1
  done
done
