#!/usr/bin/env python3
"""The handful of numbers of a bench.py JSON line one looks at first.   python tools/print_bench.py bench.json"""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d["roofline"]
print(f"headline: {d['value']:.4g} {d['unit']}, {d['ms_per_step']:.3f} ms/step, {r['bound']} frac {r['frac']:.3f}, traffic {r['traffic'] and r['traffic']['bytes'] / 1e9:.2f} GB (stale: {r['traffic'] and r['traffic']['stale']}), verified {d.get('verified', {}).get('ok')}")
if "edmdc" in d:
    e = d["edmdc"]
    print(f"gram: {e['value']:.4g} samples/s, mfma frac {e['roofline']['frac']:.3f}; k-means {e['kmeans']['kmeanspp_ms_device']:.1f} + {e['kmeans']['lloyd_ms_total']:.1f} ms")
f = d.get("edmdc_fit")
if f:
    for k in ("fit", "fit_multi"):
        print(f"{k}: {f[k]['fit_samples_per_s']:.4g} samples/s, {f[k]['wall_s']:.4f} s,", {a: round(b, 1) for a, b in f[k]["stages_ms"].items()})
    h = f["host_call"]
    print(f"host_call: fit {h['second_call_s']:.3f} s; fit_multi {h['fit_multi']['second_call_s']:.3f} s = {h['fit_multi']['samples_per_s_second_call']:.4g} samples/s, ratio {h['fit_multi']['ratio_to_device_resident_leg_plus_upload']:.3f}")
    rs = f.get("recorded_shape")
    if rs:
        print(f"recorded shape: {rs['value']:.4g} samples/s warm;", {k: (round(v['first_call_s'], 3), round(v['warm_call_s'], 4), v['max_abs_drmse_H1_10_100_vs_numpy_pinv']) for k, v in rs["pinv_options"].items() if isinstance(v, dict)},
              f"cpu {rs['cpu_baseline']['value']:.4g} samples/s")
    print("lloyd frac", f["fit"]["lloyd_roofline"] and (round(f["fit"]["lloyd_roofline"]["frac"], 3), f["fit"]["lloyd_roofline"]["stale"]), "apply frac", round(f["fit"]["roofline"]["frac"], 3))
if "config4" in d:
    print("config4:", d["config4"]["rank0_ms"], d["config4"]["verified"]["ok"])
print("cpu_baseline:", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline_reference_shape", {}).get("value"))
