#!/usr/bin/env python3
"""Per-kernel launch durations from a rocprofv3 --kernel-trace CSV, the FIRST launch of every kernel left out (code-object load and
cold caches make it an outlier: the --stats average of a profiled bench run is above bench.py's own ms_per_step for that reason alone).

    python tools/kernel_times.py run_kernel_trace.csv > kernel_times.json

{kernel: {launches, first_us, median_us, min_us, mean_us, max_us}} for the kernels that ran more than once, longest total first."""
import csv
import json
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = {}
for r in rows:
    by.setdefault(r["Kernel_Name"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {}
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    rest = v[1:] if len(v) > 1 else v
    out[k[:160]] = {"launches": len(v), "first_us": v[0], "median_us": statistics.median(rest), "min_us": min(rest),
                    "mean_us": sum(rest) / len(rest), "max_us": max(rest), "total_ms": sum(v) / 1e3}
json.dump(out, sys.stdout, indent=1)
