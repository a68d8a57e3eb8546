"""Durations of the seeding's kernels by round from a rocprofv3 --kernel-trace CSV (second seeding of tools/pp_round_profile.py)."""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for key in ("pp_round_kernel", "pp_decide_kernel"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if key in r["Kernel_Name"]]
    d = d[len(d) // 2:]                      # the second seeding
    print(key, "launches", len(d), "total ms %.1f" % (sum(d) / 1e3))
    for a, b in ((0, 8), (8, 20), (20, 50), (50, 100), (100, 200), (200, 300), (300, 400), (400, 512)):
        seg = d[a:b]
        if seg:
            print("   rounds %3d..%3d: mean %.1f us, min %.1f, max %.1f" % (a, b, statistics.mean(seg), min(seg), max(seg)))
gaps = []
pr = [r for r in rows if "pp_round_kernel" in r["Kernel_Name"] or "pp_decide_kernel" in r["Kernel_Name"]]
pr = pr[len(pr) // 2:]
for x, y in zip(pr[:-1], pr[1:]):
    gaps.append((int(y["Start_Timestamp"]) - int(x["End_Timestamp"])) / 1e3)
print("gap between consecutive seeding kernels: mean %.1f us, median %.1f us, total %.1f ms" % (statistics.mean(gaps), statistics.median(gaps), sum(gaps) / 1e3))
