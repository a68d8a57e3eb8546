#!/usr/bin/env python3
"""Print the brov:: rows of a rocprofv3 *_kernel_stats.csv (helper for profiles/)."""
import csv
import glob
import sys

path = sys.argv[1]
files = glob.glob(path + "/*/*kernel_stats.csv") if not path.endswith(".csv") else [path]
for r in csv.DictReader(open(files[0])):
    n = r["Name"].split("(")[0][:64]
    if "brov" in n:
        print("%-64s calls=%3s avg_ms=%9.3f pct=%s" % (n, r["Calls"], float(r["AverageNs"]) / 1e6, r["Percentage"]))
