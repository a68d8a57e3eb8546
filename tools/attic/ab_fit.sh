#!/bin/bash
# A/B timing of experimental library builds on the EDMDc leg of bench.py (Gram fit time, lift + gram kernels)
for lib in build_variants/libbrov2_*.so; do
  for rep in 1 2; do
    BROV2_LIBRARY=$PWD/$lib timeout -k 10 300 python bench.py --steps 1 --warmup 0 --batch 4096 --horizon 100 --no-cpu --kmeans-iters 2 --edmdc-steps 3 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())['edmdc']; print('$lib', '%.3e samples/s' % d['value'], '%.2f ms/fit' % d['ms_per_fit_gram'])" || exit 1
  done
done
