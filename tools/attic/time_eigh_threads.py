import numpy as np, time, sys
sys.path.insert(0, '.')
from threadpoolctl import threadpool_limits
from bluerov2_dynamics_amd import engine
rng = np.random.default_rng(0)
for p in (520, 532):
    G = rng.normal(size=(4000, p)); A = G.T @ G
    for nt in (1, 2, 4, 8, 16):
        with threadpool_limits(limits=nt, user_api="blas"):
            ts = []
            for _ in range(12):
                t0 = time.perf_counter(); engine.pinv_sym_host(A, 0.1); ts.append((time.perf_counter() - t0) * 1e3)
            tp = []
            for _ in range(6):
                t0 = time.perf_counter(); np.linalg.pinv(A + 0.1 * np.eye(p)); tp.append((time.perf_counter() - t0) * 1e3)
        print(f"p={p} threads={nt}: eigh min {min(ts):.1f} median {sorted(ts)[6]:.1f} max {max(ts):.1f} ms | pinv min {min(tp):.1f} median {sorted(tp)[3]:.1f} max {max(tp):.1f}", flush=True)
