#!/usr/bin/env python3
"""Board power and shader clock while a workload runs (sysfs hwmon of the amdgpu device; rocm-smi as a cross-check).

    python3 tools/power_trace.py LABEL -- <command ...>

Samples power1_average / power1_input (uW), freq1_input (gfx clock, Hz) and the power cap every 20 ms from before the command
starts until it ends, prints a summary line (idle level, busy plateau = samples above 60 % of the peak draw) and the raw
series decimated to ~50 points.  Run on the GPU box; the child is a plain subprocess (nothing here touches HIP)."""
import glob
import json
import os
import subprocess
import sys
import threading
import time


def find_hwmon():
    out = []
    for h in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        f = {k: os.path.join(h, k) for k in ("power1_average", "power1_input", "freq1_input", "power1_cap", "power1_cap_max", "temp1_input")
             if os.path.exists(os.path.join(h, k))}
        if f:
            out.append((h, f))
    return out


def rd(path):
    try:
        return float(open(path).read().strip())
    except Exception:
        return float("nan")


def main():
    label = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    hw = find_hwmon()
    if not hw:
        print(json.dumps({"label": label, "error": "no amdgpu hwmon files visible"}))
        return subprocess.call(cmd)
    samples = {h: [] for h, _ in hw}
    stop = threading.Event()

    def loop():
        while not stop.is_set():
            t = time.perf_counter()
            for h, f in hw:
                p = rd(f["power1_average"]) if "power1_average" in f else rd(f.get("power1_input", ""))
                samples[h].append((t, p / 1e6, rd(f["freq1_input"]) / 1e6 if "freq1_input" in f else float("nan"),
                                   rd(f["temp1_input"]) / 1e3 if "temp1_input" in f else float("nan")))
            time.sleep(0.02)

    th = threading.Thread(target=loop, daemon=True)
    th.start()
    time.sleep(0.5)
    t0 = time.perf_counter()
    rc = subprocess.call(cmd, stdout=subprocess.DEVNULL)
    t1 = time.perf_counter()
    time.sleep(0.3)
    stop.set()
    th.join()
    # the host shows every card of the node (other tenants' too): ours is the one whose draw follows the command -- the
    # largest rise of the mean power during the run over its level before; BROV2_POWER_ALL=1 prints all cards
    def rise(h):
        s = samples[h]
        run = [x[1] for x in s if t0 <= x[0] <= t1]
        idle = [x[1] for x in s if x[0] < t0]
        return (max(run) if run else 0.0) - (sum(idle) / len(idle) if idle else 0.0)
    if os.environ.get("BROV2_POWER_ALL") != "1":
        mine = None
        try:        # PCI address of HIP device 0, asked from a child process (this one stays off the GPU)
            q = subprocess.run([sys.executable, "-c", "import torch; p = torch.cuda.get_device_properties(0); "
                                "print('%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id))"],
                               capture_output=True, text=True, timeout=120).stdout.strip().splitlines()[-1]
            mine = [e for e in hw if os.path.basename(os.path.realpath(os.path.join(e[0], "..", ".."))) == q]
        except Exception:
            mine = None
        hw = mine if mine else sorted(hw, key=lambda e: -rise(e[0]))[:1]
    for h, f in hw:
        s = samples[h]
        run = [x for x in s if t0 <= x[0] <= t1]
        if not run:
            continue
        pk = max(x[1] for x in run)
        if not pk > 0:
            continue
        busy = [x for x in run if x[1] >= 0.6 * pk]
        idle = [x for x in s if x[0] < t0]
        dec = run[:: max(1, len(run) // 50)]
        print(json.dumps({
            "label": label, "hwmon": h, "rc": rc, "wall_s": t1 - t0, "samples": len(run),
            "power_cap_W": rd(f["power1_cap"]) / 1e6 if "power1_cap" in f else None,
            "idle_W": sum(x[1] for x in idle) / max(len(idle), 1), "idle_MHz": sum(x[2] for x in idle) / max(len(idle), 1),
            "peak_W": pk, "busy_mean_W": sum(x[1] for x in busy) / len(busy), "busy_mean_MHz": sum(x[2] for x in busy) / len(busy),
            "busy_min_MHz": min(x[2] for x in busy), "busy_max_MHz": max(x[2] for x in busy), "busy_samples": len(busy),
            "temp_C_max": max(x[3] for x in run),
            "series_t_W_MHz": [(round(x[0] - t0, 2), round(x[1], 1), round(x[2])) for x in dec]}))
    return rc


if __name__ == "__main__":
    sys.exit(main())
