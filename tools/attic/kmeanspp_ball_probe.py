"""Would a two-level screening pay in the k-means++ seeding?  Today a round reads a float copy of every sample (48 B) + `closest` (8 B) to
find that most samples are out of reach of the round's nine points (tools/kmeanspp_screen_probe.py: 89-98 % of the 16-sample rows).
On trajectory-ordered data 16 consecutive samples are neighbours: a ball (centre, radius) per row costs 4 B per sample, and
    d(point, ball centre) - radius >= max over the row of sqrt(closest)
certifies the whole row without touching its samples.  This probe counts, per round, the rows the ball test certifies against the rows
the per-sample test certifies (config-3 data, the shipped seeding's own centres in order).
    python3 tools/kmeanspp_ball_probe.py [pairs] [rowlen]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bluerov2_dynamics_amd import _lib, engine

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
RL = int(sys.argv[2]) if len(sys.argv) > 2 else 16
shuffle = len(sys.argv) > 3 and sys.argv[3] == "shuffle"
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
if shuffle:
    X = X[torch.randperm(X.shape[0], device=dev, generator=g)]
N = X.shape[0]
mean = X.mean(0)
C, idx = engine.kmeanspp_dev(X, k, mean=mean.cpu().numpy(), random_state=0, ctx=ctx)
Xc = (X - mean).float()
Cf = C.float()
nr = N // RL
Xr = Xc[: nr * RL].view(nr, RL, n)
ctr = Xr.mean(1)                                                   # ball centres
rad = (Xr - ctr[:, None, :]).norm(dim=2).max(1).values             # radii
print(f"{N} samples, rows of {RL}: median radius {float(rad.median()):.4f}, 90 % {float(rad.quantile(0.9)):.4f}, max {float(rad.max()):.4f}", flush=True)
x2 = (Xc * Xc).sum(1)
def dist2(P):
    return (x2[:, None] - 2 * Xc @ P.T + (P * P).sum(1)[None]).clamp_min(0)
closest = dist2(Cf[:1])[:, 0]
for c in range(1, k):
    if c in (2, 5, 8, 10, 20, 50, 100, 200, 300, 400, 500):
        cand = torch.multinomial((closest / closest.sum()).double(), 8, replacement=True, generator=g)
        P = torch.cat([Xc[cand], Cf[c - 1:c]])
        D = dist2(P)
        near = (D < closest[:, None] * (1 + 1e-5) + 1e-9).any(1)
        rows_sample = near[: nr * RL].view(nr, RL).any(1)
        sq = closest[: nr * RL].sqrt().view(nr, RL).max(1).values
        dc = torch.cdist(ctr, P)                                    # [rows, 9]
        ball_ok = ((dc - rad[:, None]) >= sq[:, None] * (1 + 1e-5) + 1e-6).all(1)
        print(f"round {c:3d}: rows needing work by the per-sample test {float(rows_sample.double().mean()):.4f}; rows NOT certified by the ball test "
              f"{float((~ball_ok).double().mean()):.4f}; median sqrt(closest) {float(closest.sqrt().median()):.3f}", flush=True)
    closest = torch.minimum(closest, dist2(Cf[c:c + 1])[:, 0])
