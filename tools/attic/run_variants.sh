#!/bin/bash
# GPU box: time the headline rollout launch with every library under build_variants/ (and the shipped one).
#   tools/run_variants.sh [extra bench.py flags]   -> gpurun_out/variants.txt
#   VARIANTS="a b c" restricts the run to build_variants/{a,b,c}; REPS=n repeats every library n times (interleaved: A/B on one box)
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/variants.txt
: > $out
libs=bluerov2_dynamics_amd/libbrov2.so
if [ -n "$VARIANTS" ]; then for v in $VARIANTS; do libs="$libs build_variants/$v/libbrov2.so"; done
else for l in build_variants/*/libbrov2.so; do case $l in *host_sanitizer*|*experiments*) ;; *) libs="$libs $l";; esac; done; fi
for rep in $(seq 1 ${REPS:-1}); do
for lib in $libs; do
  [ -f "$lib" ] || continue
  name=$(basename $(dirname $lib))
  BROV2_LIBRARY=$PWD/$lib timeout -k 10 300 python3 bench.py --no-edmdc --no-cfg4 --no-cpu --no-ar1 --steps 4 --warmup 1 "$@" > gpurun_out/var_$name.json 2> gpurun_out/var_$name.err
  rc=$?
  python3 - "$name" "$rc" gpurun_out/var_$name.json >> $out <<'PY'
import json, sys
name, rc, path = sys.argv[1:4]
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    v = d.get("verified", {})
    print(f"{name:28s} rc={rc} kernel_ms={d['roofline']['kernel_ms']:.3f} each={['%.3f' % x for x in d['roofline']['kernel_ms_each']]} "
          f"steps/s={d['value']:.4e} verified_ok={v.get('ok')} max_rel_err={v.get('max_rel_err')}")
except Exception as e:
    print(f"{name:28s} rc={rc} FAILED {e}")
PY
  tail -1 $out
  [ $rc -eq 124 ] && { echo "timeout: stopping" >> $out; exit 1; }
done
done
exit 0
