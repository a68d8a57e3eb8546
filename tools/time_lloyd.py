"""Lloyd on BASELINE config-3 data (1e7 trajectory-ordered states, k = 512): candidate-filtered E-step against the full scan,
fixed iteration count, same seeds.   python3 tools/time_lloyd.py [pairs] [iters] [all|default]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bluerov2_dynamics_amd import _lib, engine

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
ctx = _lib.default_context(0)
n, r, k, L = 12, 8, 512, 500
nb = max(1, pairs // L)
Ue = torch.empty((nb, L, r), dtype=torch.float64, device=dev)
engine.fill_controls_dev(Ue, "btu", "ar1", seed=0xED3D, b0=0, T_total=L, ctx=ctx)
Xe = torch.empty((nb, L + 1, n), dtype=torch.float64, device=dev)
engine.rollout_dev(_lib.THRUSTER_EULER, "euler", torch.zeros((nb, n), dtype=torch.float64, device=dev), Ue, 0.02, traj=Xe, layout="btu", ctx=ctx)
g = torch.Generator(device=dev); g.manual_seed(1234)
sig = torch.tensor([5e-4] * 3 + [1e-3] * 3 + [5e-4] * 3 + [1e-3] * 3, dtype=torch.float64, device=dev)
Xe += torch.randn(Xe.shape, generator=g, dtype=torch.float64, device=dev) * sig
X = Xe.view(-1, n)
res = {}
runs = (("filtered", 0), ("full scan", 1), ("filtered", 0))
if len(sys.argv) > 3 and sys.argv[3] == "default":        # the shipped loop only (counter runs: every launch of the kernel is a filtered one but the first)
    runs = (("filtered", int(os.environ.get("BROV2_KMEANS_VARIANT", "0"))),)
if len(sys.argv) > 3 and sys.argv[3] == "all":            # + the caller's order, bounds off, and (experiments build: BROV2_LIBRARY=build_variants/experiments/libbrov2.so) the scalar-record kernel
    runs += (("filtered, caller's order", 2), ("distance bounds off", 4))
    if _lib.load_library().brov_experiments_build():
        runs += (("scalar records: filtered", 256), ("scalar records: full scan", 257), ("scalar records: caller's order", 258))
for name, v in runs:
    ctx.set_kmeans_variant(v)
    tm = {}
    ctx.set_timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    C, inertia, n_iter = engine.kmeans_centers_dev(X, k, random_state=0, max_iter=iters, ctx=ctx, timings=tm)
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
    ctx.set_timing(False)
    print(f"{name:32s}: {n_iter} iterations, seeding {tm['kmeanspp_ms']:.1f} ms, Lloyd {tm['lloyd_ms']:.1f} ms = {tm['lloyd_ms'] / (n_iter + 1):.3f} ms per E+M step, "
          f"wall {wall * 1e3:.0f} ms, inertia {inertia:.9e}", flush=True)
    res[name] = C.cpu().numpy()
if "full scan" in res:
    print("centres filtered vs full scan: max rel diff", float(np.max(np.abs(res["filtered"] - res["full scan"]) / np.maximum(1, np.abs(res["full scan"])))))
