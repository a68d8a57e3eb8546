"""Host path of the Gram at 1e7 rows (NumPy arrays in): how much of it is the upload?  Run on the GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bluerov2_dynamics_amd import engine, _lib
rng = np.random.default_rng(0)
N, n, r, k = 10_000_000, 12, 8, 512
X = rng.normal(0, 0.3, (N, n)); U = rng.uniform(-1, 1, (N, r)); C = rng.normal(0, 0.3, (k, n))
for rep in range(3):
    t0 = time.perf_counter()
    GtG, GtY, npairs = engine.gram([X], [U], C, 1.0)
    print(f"host-path gram, N = {N}: {time.perf_counter() - t0:.3f} s (device lift + Gram alone: 0.135 s)")
import torch
t0 = time.perf_counter(); Xd = torch.from_numpy(X).cuda(); Ud = torch.from_numpy(U).cuda(); torch.cuda.synchronize()
print(f"torch pageable upload of X, U (1.6 GB): {time.perf_counter() - t0:.3f} s")
