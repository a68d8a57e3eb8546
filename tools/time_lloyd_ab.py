"""A/B of Lloyd variants on BASELINE config-3 data (1e7 states, k = 512), 300 iterations each:
    python3 tools/time_lloyd_ab.py [iters] v1 v2 ...      (variants of edmdc_set_kmeans_variant; default 0 16)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bluerov2_dynamics_amd import _lib, engine

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
variants = [int(v) for v in sys.argv[2:]] or [0, 16]
pairs = 10_000_000
dev = torch.device("cuda", 0)
ctx = _lib.default_context(0)
n, r, k, L = 12, 8, int(os.environ.get("AB_K", "512")), 500
nb = max(1, pairs // L)
Ue = torch.empty((nb, L, r), dtype=torch.float64, device=dev)
engine.fill_controls_dev(Ue, "btu", "ar1", seed=0xED3D, b0=0, T_total=L, ctx=ctx)
Xe = torch.empty((nb, L + 1, n), dtype=torch.float64, device=dev)
engine.rollout_dev(_lib.THRUSTER_EULER, "euler", torch.zeros((nb, n), dtype=torch.float64, device=dev), Ue, 0.02, traj=Xe, layout="btu", ctx=ctx)
g = torch.Generator(device=dev); g.manual_seed(1234)
sig = torch.tensor([5e-4] * 3 + [1e-3] * 3 + [5e-4] * 3 + [1e-3] * 3, dtype=torch.float64, device=dev)
Xe += torch.randn(Xe.shape, generator=g, dtype=torch.float64, device=dev) * sig
X = Xe.view(-1, n)
if os.environ.get("AB_PAD"):          # experiment: rows padded to a stride of 16 doubles (one 128-byte line per gathered row)
    Xp = torch.zeros((X.shape[0], 16), dtype=torch.float64, device=dev)
    Xp[:, :n] = X
    X = Xp[:, :n]
ref = None
for rep in range(2):
    for v in variants:
        ctx.set_kmeans_variant(v)
        tm = {}
        ctx.set_timing(True)
        C, inertia, n_iter = engine.kmeans_centers_dev(X, k, random_state=0, max_iter=iters, ctx=ctx, timings=tm)
        torch.cuda.synchronize()
        ctx.set_timing(False)
        Ch = C.cpu().numpy()
        same = "" if ref is None else f", centres == first run: {np.array_equal(Ch, ref)}"
        ref = Ch if ref is None else ref
        print(f"variant {v:2d}: {n_iter} iterations, Lloyd {tm['lloyd_ms']:.1f} ms = {tm['lloyd_ms'] / (n_iter + 1):.3f} ms per E+M step, seeding {tm['kmeanspp_ms']:.1f} ms{same}", flush=True)
