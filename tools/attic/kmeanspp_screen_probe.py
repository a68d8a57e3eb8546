"""How many 16-sample rows of a k-means++ round need their fp64 coordinates?  A row needs them only if for some sample in it one
of the round's points (8 candidates drawn in proportion to `closest`, and the centre chosen last) is nearer than the sample's
nearest chosen centre so far -- everything else contributes min(closest, d) = closest, which a float copy of the coordinates can
certify.  Emulates rounds c of the seeding of BASELINE config-3 data (centres = the shipped seeding's own, in order).
    python3 tools/kmeanspp_screen_probe.py [pairs]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bluerov2_dynamics_amd import _lib, engine

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
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
N = X.shape[0]
mean = X.mean(0)
C, idx = engine.kmeanspp_dev(X, k, mean=mean.cpu().numpy(), random_state=0, ctx=ctx)
Xc = X - mean
x2 = (Xc * Xc).sum(1)
def dist2(P):                                          # [N, m] squared distances to the points P [m, n]
    return (x2[:, None] - 2 * Xc @ P.T + (P * P).sum(1)[None]).clamp_min(0)
closest = dist2(C[:1])[:, 0]
tot_rows = 0.0; tot_w = 0.0
for c in range(1, k):
    if c in (2, 5, 10, 20, 50, 100, 200, 300, 400, 500):
        cand = torch.multinomial(closest / closest.sum(), 8, replacement=True, generator=g)
        P = torch.cat([Xc[cand], C[c - 1:c]])          # 8 candidates + the centre chosen last (its update is still owed)
        D = dist2(P)
        near = (D < closest[:, None] * (1 + 1e-5) + 1e-9).any(1)
        nr = N // 16
        rows = near[: nr * 16].view(nr, 16).any(1)
        print(f"round {c:3d}: samples nearer to one of the 9 points {float(near.double().mean()):.4f}, rows of 16 needing fp64 {float(rows.double().mean()):.4f}", flush=True)
    closest = torch.minimum(closest, dist2(C[c:c + 1])[:, 0])
