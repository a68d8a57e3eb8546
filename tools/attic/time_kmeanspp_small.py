"""k-means++ seeding at small sample counts: one cooperative launch against two launches per centre (kernel ms)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bluerov2_dynamics_amd import _lib, engine
rng = np.random.default_rng(0)
for N, k in ((300, 100), (4000, 500), (36658, 500), (300000, 500), (2000000, 512)):
    X = torch.from_numpy(np.cumsum(rng.normal(0, 0.05, (N, 12)), 0)).cuda()
    for name, v in (("two launches per centre", 0),):
        ctx = _lib.Context(0); ctx.set_timing(True)
        ts = []
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            engine.kmeanspp_dev(X, k, mean=None, random_state=0, ctx=ctx)
            torch.cuda.synchronize(); ts.append(((time.perf_counter() - t0) * 1e3, ctx.last_kernel_ms()))
        print(f"N={N:8d} k={k}: {name:15s} wall {min(t[0] for t in ts):7.2f} ms, kernels {min(t[1] for t in ts):7.2f} ms = {min(t[1] for t in ts) / k * 1e3:6.1f} us per centre", flush=True)
        ctx.close()
