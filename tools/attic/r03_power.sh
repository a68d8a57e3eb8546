#!/bin/bash
# GPU box: board power / shader clock of (a) the config-2 rollout launch repeated for ~4 s, (b) its endpoint-only form (no
# trajectory stores), (c) the EDMDc Gram, (d) a pure v_fma_f64 stream (tools/ubench_power, no memory traffic).
set -e -o pipefail
out=gpurun_out/r03_power; mkdir -p $out
hipcc --offload-arch=gfx950 -O3 tools/ubench_power.hip -o /tmp/ubench_power 2>/dev/null
ls /sys/class/drm/card*/device/hwmon/hwmon*/ > $out/hwmon_files.txt 2>&1 || true
python3 tools/power_trace.py rollout_rk4_stored -- python3 bench.py --steps 300 --warmup 3 --no-edmdc --no-cfg4 --no-cpu --no-variants --no-ar1 > $out/rollout.json
python3 tools/power_trace.py rollout_rk4_endpoint_only -- python3 bench.py --steps 300 --warmup 3 --no-store --no-edmdc --no-cfg4 --no-cpu --no-variants --no-ar1 > $out/rollout_nostore.json
python3 tools/power_trace.py edmdc_gram -- python3 tools/time_fit.py 10000000 12 > $out/fit.json
python3 tools/power_trace.py fma_stream -- /tmp/ubench_power > $out/fma.json
rocm-smi --showpower --showclocks --showmaxpower > $out/rocm_smi_idle.txt 2>&1 || true
for f in rollout rollout_nostore fit fma; do python3 - $out/$f.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: d = json.loads(l)
    except Exception: continue
    print({k: d[k] for k in d if k != "series_t_W_MHz"})
PY
done
