#!/bin/bash
# tools/kstats_run.sh OUTDIR -- python3 script args...   : rocprofv3 --kernel-trace --stats of a command, top kernels printed (GPU box)
out=$1; shift; shift
root=$(pwd)
mkdir -p $out
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/trace -o run -- "$@" > $root/$out/trace.log 2>&1 )
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
python3 - $out <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] + "/kernel_stats.csv")))
for r in rows[:16]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} total_ms {float(r["TotalDurationNs"]) / 1e6:9.2f} avg_us {float(r["AverageNs"]) / 1e3:9.1f} min_us {float(r["MinNs"]) / 1e3:9.1f}')
PY
