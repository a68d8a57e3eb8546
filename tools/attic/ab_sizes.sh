#!/bin/bash
# rollout leg at several batch sizes, with and without trajectory storage
for args in "--batch 65536" "--batch 65536 --no-store" "--batch 131072" "--batch 131072 --no-store" "--batch 262144 --horizon 2000" "--batch 262144 --horizon 2000 --no-store"; do
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-edmdc $args 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$args', '%.3e steps/s' % d['value'], '%.3f ms' % d['roofline']['kernel_ms'], 'HBM alg %.0f GB/s' % d['roofline']['hbm']['achieved'])" || exit 1
done
