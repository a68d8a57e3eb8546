"""k-means++ seeding (edmdc_kmeanspp_dev) at the benchmark size: 10.02e6 rows x 12, k = 512.  Run on the GPU box."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bluerov2_dynamics_amd import engine
rng = np.random.default_rng(0)
N, n, k = 10_020_000, 12, 512
X = torch.from_numpy(rng.normal(0, 1, (N, n))).cuda()
ctx = engine.default_context(0)
ctx.set_timing(True)
for rep in range(2):
    C, idx = engine.kmeanspp_dev(X, k, mean=X.mean(0).cpu().numpy(), random_state=0, ctx=ctx)
    print("kmeanspp ms", ctx.last_kernel_ms())

tm = {}
Ck, inertia, n_iter = engine.kmeans_centers_dev(X, k, random_state=0, max_iter=10, ctx=ctx, timings=tm)
print("lloyd ms per iteration", tm["lloyd_ms"] / n_iter, "iterations", n_iter, "inertia", inertia)
