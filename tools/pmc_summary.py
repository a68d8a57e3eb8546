#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc passes (one directory per pass, as written by tools/pmc_pass.sh) into per-kernel means.

    python tools/pmc_summary.py gpurun_out/pmc_rX [kernel-substring ...]  > summary.json

Values are summed over the dispatch's counter instances by rocprofv3 already (one row per dispatch and counter);
this script averages over dispatches of kernels whose name contains the substring.  FETCH_SIZE is doubled on
gfx950 per /opt/skills/guides/MI355X_MICROARCH.md (the counter tallies 128-B requests as 64 B); both are in KiB.
"""
import csv
import glob
import json
import sys

root = sys.argv[1]
pats = sys.argv[2:] or ["rollout_kernel", "gram_kernel", "lift_rows_kernel", "kmeans_assign_kernel", "propagate_kernel"]
out = {p: {} for p in pats}
for f in glob.glob(root + "/*/**/*counter_collection.csv", recursive=True):
    acc = {}
    for r in csv.DictReader(open(f)):
        for p in pats:
            if p in r["Kernel_Name"]:
                acc.setdefault((p, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (p, c), v in acc.items():
        out[p][c] = sum(v) / len(v)
        out[p]["launches_" + c] = len(v)
for p, d in out.items():
    if "FETCH_SIZE" in d:
        d["hbm_read_GB_per_launch_corrected_x2"] = d["FETCH_SIZE"] * 1024 * 2 / 1e9
    if "WRITE_SIZE" in d:
        d["hbm_write_GB_per_launch"] = d["WRITE_SIZE"] * 1024 / 1e9
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_total_GB_per_launch"] = d["hbm_read_GB_per_launch_corrected_x2"] + d["hbm_write_GB_per_launch"]
json.dump(out, sys.stdout, indent=1)
