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
NAMES = {"rollout_pair_kernel<1, 2,": "rollout", "rollout_pair_kernel<1, 0,": "rollout_btu", "fill_ar1_btu_kernel": "fill_ar1_btu",     # thruster model, RK4, paired time-major layout: the benchmark kernel
          "gram_kernel": "gram", "lift_rows_kernel": "lift", "kmeans_assign_kernel": "kmeans_assign_scalar_records", "kmeans_assign_lds_kernel": "kmeans_assign",
         "propagate_kernel": "propagate", "pp_round_kernel": "kmeanspp_round", "pp_decide_kernel": "kmeanspp_decide", "lift_tail_kernel": "lift_tail",
         "gram_kernel<false>": "gram", "gram_kernel<true>": "wty_gram", "wrows_kernel": "wrows", "rows_times_pt_simple_kernel": "wrows_simple"}
NAMES.pop("gram_kernel")
pats = sys.argv[2:] or [p_ for p_ in NAMES if p_ not in ("rollout_pair_kernel<1, 0,", "fill_ar1_btu_kernel")]
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
for p, d in out.items():
    if d.get("SQ_INSTS_MFMA"):
        # SQ_VALU_MFMA_BUSY_CYCLES counts per SIMD; GRBM_GUI_ACTIVE is summed over the 8 XCDs by rocprofv3
        d["cycles_per_mfma"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_INSTS_MFMA"]
        if d.get("GRBM_GUI_ACTIVE"):
            d["mfma_busy_fraction"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
out = {NAMES.get(p, p): d for p, d in out.items()}
out["_method"] = ("rocprofv3 --pmc <group> --kernel-trace, one pass per counter group (tools/profile_round.sh), python3 bench.py --steps 1 "
                  "--warmup 1 --edmdc-steps 1 --no-cpu; per-kernel means over the dispatches of each pass; FETCH_SIZE doubled per "
                  "MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); sizes in KiB; SQ_*_CYCLES in units of 4 clocks")
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
try:
    from bench import kernel_source_sha
    out["_kernel_source_sha"] = kernel_source_sha()       # bench.py prints it next to the traffic figure: stale summaries show
except Exception as e:                                    # noqa: BLE001
    out["_kernel_source_sha"] = None
json.dump(out, sys.stdout, indent=1)
