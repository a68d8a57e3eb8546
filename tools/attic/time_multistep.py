#!/usr/bin/env python3
"""Timing of the f1 path (edmdc_multistep_se) at the reference's recorded size: 45 823 samples, H = 100, k = 512."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bluerov2_dynamics_amd import engine, _lib

n, r, k, N, H = 12, 8, 512, 45823, 100
rng = np.random.default_rng(0)
X = rng.normal(size=(N, n)) * 0.3
U = rng.uniform(-1, 1, (N, r))
C = rng.normal(size=(k, n)) * 0.3
d = n + k
A = 0.95 * np.eye(d) + rng.normal(size=(d, d)) * 1e-3
B = rng.normal(size=(d, r)) * 1e-2
ctx = _lib.Context(0)
ctx.set_timing(True)
for rep in range(3):
    t0 = time.perf_counter()
    se, _ = engine.multistep_se(X, U, C, 1.0, A, B, H, ctx=ctx)
    wall = time.perf_counter() - t0
    kms = ctx.last_kernel_ms()
    print("TB=%s rep %d: kernel %.2f ms (%.1f TFLOP/s) wall %.1f ms se=%.6e" % (
        os.environ.get("BROV2_PROP_TB", "default"), rep, kms, 2.0 * (N - H) * d * (d + r) * H / (kms * 1e-3) / 1e12, wall * 1e3, se))
