#!/usr/bin/env python3
"""The reference's 4-model comparison (training/train_tank_brov2_full_comparison.py: main :894-1048) on the MI355X
engine, for the rows this engine covers: Koopman EDMDc, Fossen (BlueROV2) and the learned double integrator.
The PINc row is a PyTorch network of the reference that runs unchanged on PyTorch-ROCm; it is not part of this repo: its
three RMSEs (from the reference's own multistep_rmse_endpoint_pinc) can be handed in with --pinc-row to complete the table
and the ranking (tests/golden/cfg5_pinc.npz holds them for the CSV fixture).

    python examples/full_comparison.py path/to/koopman_dataset_50Hz.csv [--rbfs 500 --gamma 3 --ridge 0.1 --rk4]
    python examples/full_comparison.py path/to/koopman_dataset_50Hz_with_wrench.csv --variant wrench   # train_tank_brov2_wrench_comp.py
    python examples/full_comparison.py path/to/koopman_dataset_50Hz_with_wrench.csv --variant quat     # train_tank_brov2_wrench_quat.py

Same data handling (load_dataset, 80/20 split), same metrics (endpoint RMSE at H = 1/10/100 over all sliding
windows, one vehicle object for all windows => thruster lag carried across windows), same table layout.
"""
import argparse
import os
import sys
from time import perf_counter

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from bluerov2_dynamics_amd.baselines import DoubleIntegrator          # noqa: E402
from bluerov2_dynamics_amd.data import load_dataset                   # noqa: E402
from bluerov2_dynamics_amd.fossen.BlueROV2 import BlueROV2             # noqa: E402
from bluerov2_dynamics_amd.fossen.BlueROV2_thrust import BlueROV2 as BlueROV2Wrench          # noqa: E402
from bluerov2_dynamics_amd.fossen.BlueROV2_wrench import BlueROV2 as BlueROV2Quat            # noqa: E402
from bluerov2_dynamics_amd.Koopman.koopmanEDMDc import KoopmanEDMDc    # noqa: E402

TRAIN_SPLIT = 0.80


ROWS = ("Koopman", "Fossen (BlueROV2)", "Double Integrator", "PINc (ResDNN)")


def compare(csv_path, n_rbfs=500, gamma=3.0, ridge=1e-1, integrator="euler", centers=None, verbose=True, variant="thruster", pinc_row=None):
    """Returns dict(table [3,3] rows Koopman / Fossen / DI x H = 1, 10, 100, timings, dt, split); with pinc_row (the
    reference network's three RMSEs, computed elsewhere) the table has the reference's four rows and `ranking` [4,3] gives
    each model's rank per horizon (0 = best), training/train_tank_brov2_full_comparison.py:996-1001.
    variant: "thruster" (8 PWM inputs, Euler angles), "wrench" (6-D body wrench, Euler angles), "quat" (wrench, quaternion
    state; RK4 exists only for the thruster script in the reference)."""
    X, U, dt = load_dataset(csv_path, verbose=verbose, variant=variant)
    nx, nu = X.shape[1], U.shape[1]
    make_rov = {"thruster": lambda: BlueROV2(dt=dt), "wrench": BlueROV2Wrench, "quat": BlueROV2Quat}[variant]
    if len(X) < 3:
        raise RuntimeError("Not enough samples to train/evaluate.")
    split = int(TRAIN_SPLIT * len(X))
    Xtr, Utr, Xte, Ute = X[:split], U[:split], X[split:], U[split:]
    t = {}
    t0 = perf_counter()
    koop = KoopmanEDMDc(state_dim=nx, input_dim=nu, n_rbfs=n_rbfs, gamma=gamma, ridge=ridge)
    koop.fit(Xtr, Utr, centers=centers)
    t["fit_koopman"] = perf_counter() - t0
    t0 = perf_counter()
    di = DoubleIntegrator.fit(Xtr, Utr, dt, ridge=1e-3, quaternion=(variant == "quat"))
    t["fit_di"] = perf_counter() - t0
    rows = []
    for name, fn in (("Koopman", lambda H: koop.multistep_rmse(Xte, Ute, H=H)),
                     ("Fossen (BlueROV2)", lambda H: make_rov().multistep_rmse_endpoint(Xte, Ute, H, dt, integrator)),
                     ("Double Integrator", lambda H: di.multistep_rmse_endpoint(Xte, Ute, H, dt, integrator))):
        vals = []
        for H in (1, 10, 100):
            t0 = perf_counter()
            vals.append(fn(H))
            t[f"{name}_H{H}"] = perf_counter() - t0
        rows.append(vals)
    if pinc_row is not None:
        rows.append([float(v) for v in pinc_row])
    table = np.array(rows)
    if verbose:
        print(f"\n[metrics] Endpoint RMSE (full {nx}D state) with identical evaluator:")
        print("  Model                 | 1-step RMSE | 10-step RMSE | 100-step RMSE")
        print("  ----------------------|------------:|-------------:|--------------:")
        for name, r in zip(ROWS, table):
            print(f"  {name:<21s} | {r[0]:11.6f} | {r[1]:12.6f} | {r[2]:13.6f}")
        print("\n[timing] seconds:", {k: round(v, 4) for k, v in t.items()})
    return dict(table=table, timings=t, dt=dt, split=split, model=koop, ranking=np.argsort(np.argsort(table, axis=0), axis=0), rows=ROWS[:len(table)])


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--rbfs", type=int, default=500)
    ap.add_argument("--gamma", type=float, default=3.0)
    ap.add_argument("--ridge", type=float, default=1e-1)
    ap.add_argument("--rk4", action="store_true")
    ap.add_argument("--variant", default="thruster", choices=["thruster", "wrench", "quat"])
    ap.add_argument("--pinc-row", type=float, nargs=3, default=None, metavar=("RMSE1", "RMSE10", "RMSE100"),
                    help="the reference PINc network's endpoint RMSEs on the same test split (completes the table)")
    a = ap.parse_args()
    compare(a.csv, a.rbfs, a.gamma, a.ridge, "rk4" if a.rk4 else "euler", variant=a.variant, pinc_row=a.pinc_row)
