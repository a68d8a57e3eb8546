"""Where the time of a small KoopmanEDMDc.fit goes (the reference's recorded-data size and smaller).  Run on the GPU box."""
import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bluerov2_dynamics_amd.Koopman.koopmanEDMDc import KoopmanEDMDc
rng = np.random.default_rng(0)
for N in (1600, 36658):
    X = np.cumsum(rng.normal(0, 0.02, (N, 12)), 0)
    U = rng.uniform(-1, 1, (N, 8))
    for rep in range(3):
        m = KoopmanEDMDc(state_dim=12, input_dim=8, n_rbfs=500, gamma=3.0, ridge=0.1)
        t0 = time.perf_counter(); m.fit(X, U); t1 = time.perf_counter()
        r = m.multistep_rmse(X, U, H=100); t2 = time.perf_counter()
        print(f"N={N} rep {rep}: fit {t1 - t0:.4f} s, multistep_rmse(H=100) {t2 - t1:.4f} s")
    m = KoopmanEDMDc(state_dim=12, input_dim=8, n_rbfs=500, gamma=3.0, ridge=0.1)
    pr = cProfile.Profile(); pr.enable(); m.fit(X, U); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
