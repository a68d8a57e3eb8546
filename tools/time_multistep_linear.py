"""KoopmanEDMDc.multistep_rmse at the reference's recorded size (45 823 samples, 500 RBFs, H = 100; training/best_results.txt:801 logs 41.19 s):
the default H-step propagation against method="linear" (one pass, explicit powers of A).  GPU box: python3 tools/time_multistep_linear.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
from bluerov2_dynamics_amd import _lib, engine  # noqa: E402
from bluerov2_dynamics_amd.Koopman.koopmanEDMDc import KoopmanEDMDc  # noqa: E402

N, n, r, k = 45823, 12, 8, 500
rng = np.random.default_rng(0)
X = np.cumsum(rng.normal(0, 0.01, (N, n)), 0)
U = rng.uniform(-1, 1, (N, r))
m = KoopmanEDMDc(state_dim=n, input_dim=r, n_rbfs=k, gamma=3.0, ridge=0.1)
m.fit(X, U)
ctx = _lib.default_context()
for H in (1, 10, 100):
    for method in ("propagate", "linear"):
        m.multistep_rmse(X, U, H, method=method)
        ctx.set_timing(True)
        t0 = time.perf_counter()
        v = m.multistep_rmse(X, U, H, method=method)
        wall = time.perf_counter() - t0
        kms = ctx.last_kernel_ms()
        ctx.set_timing(False)
        t0 = time.perf_counter()
        if method == "linear":
            engine.linear_coefficients(m.A_, m.B_, n, H)
        coef = time.perf_counter() - t0
        print(f"H = {H:3d} {method:9s}: rmse {v:.12e}  wall {wall * 1e3:7.2f} ms  kernels {kms:7.3f} ms  host coefficients {coef * 1e3:6.2f} ms")
